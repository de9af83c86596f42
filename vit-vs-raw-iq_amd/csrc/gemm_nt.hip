// bf16 MFMA GEMM  C[M,N] = epilogue(A[M,K] * B[N,K]^T), fp32 accumulate (gfx950).
//
// Replaces every nn.Linear forward on the path (multi_head_attention.py:18,28;
// position_wise_feed_forward.py:13-16), the non-overlapping conv embeddings
// (V/.../patch_embedding.py:9-15, R/.../patch_embedding.py:29-43) and -- with B pointing at the
// transposed weight shadow -- their data gradients.
//
// Shapes on this path: M = frames*tokens (1e4..1e5), N,K in 128..3072.  Most instances are
// HBM-bound (K small), so the design goals are: A streamed once per N-tile through L2 (tiles that
// share an A row-block are consecutive after the XCD remap), every global access a 16 B/lane
// vector on 128 B rows, the whole elementwise tail (bias, ReLU, positional add, dropout, gate,
// residual) fused into the epilogue, and bf16 rows stored as contiguous 128 B segments.
//
// Tile: 128 x BN (BN = 128 | 64), BK = 64, 4 waves (2x2), mfma_f32_16x16x32_bf16.
// LDS tile rows are 128 B; 16 B chunk c of row r lives at chunk c ^ ((r>>1)&7), which makes the
// ds_read_b128 fragment reads (16 rows x one chunk column per lane group) bank-conflict free.
// Epilogue: register-only (see gemm_epilogue): C^T accumulators + v_permlane16_swap give each lane 8
// consecutive columns of a row.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int BM = 128, BK = 64, GEMM_THREADS = 256;


__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int BN, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(const GemmParams p) {
  constexpr int WN = BN / 2;        // wave tile columns
  constexpr int NT = WN / 16;       // 16-col MFMA tiles per wave
  constexpr int MT = 4;             // 16-row MFMA tiles per wave (wave tile rows = 64)
  constexpr int A_CH = BM * 8 / GEMM_THREADS;  // 16 B chunks per thread
  constexpr int B_CH = BN * 8 / GEMM_THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* As = reinterpret_cast<bf16*>(smem);                 // [BM][64]
  bf16* Bs = reinterpret_cast<bf16*>(smem + BM * BK * 2);   // [BN][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[A_CH], rb[B_CH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      const int id = tid + c * GEMM_THREADS, row = id >> 3, kc = id & 7;
      const int gm = m0 + row, gk = k0 + kc * 8;
      bf16x8 v = {};
      if (gm < p.M && gk < p.K) v = *reinterpret_cast<const bf16x8*>(p.A + (long)gm * p.lda + gk);
      ra[c] = v;
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      const int id = tid + c * GEMM_THREADS, row = id >> 3, kc = id & 7;
      const int gn = n0 + row, gk = k0 + kc * 8;
      bf16x8 v = {};
      if (gn < p.N && gk < p.K) v = *reinterpret_cast<const bf16x8*>(p.B + (long)gn * p.ldb + gk);
      rb[c] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      const int id = tid + c * GEMM_THREADS, row = id >> 3, kc = id & 7;
      *reinterpret_cast<bf16x8*>(As + row * BK + swz(row, kc) * 8) = ra[c];
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      const int id = tid + c * GEMM_THREADS, row = id >> 3, kc = id & 7;
      *reinterpret_cast<bf16x8*>(Bs + row * BK + swz(row, kc) * 8) = rb[c];
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MT], bfr[NT];
      const int ch = s * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * 64 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK + swz(row, ch) * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK + swz(row, ch) * 8);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: see gemm_epilogue
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  gemm_epilogue<4, BN / 32, EPI>(p, acc, m0 + (wave >> 1) * 64, n0 + (wave & 1) * (BN / 2), lane);
}


// ---------------------------------------------------------------------------------------------
// Fast path (K % 32 == 0): operands stream HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging)
// through a 3-slot ring, two 32-deep K stages in flight behind the one being multiplied, counted
// s_waitcnt vmcnt + raw s_barrier (guide 5, "Pipelining across barriers").  The register-staged
// kernel above keeps one stage in flight and exposes the full HBM latency on every K step
// (measured 2.0-2.9 TB/s algorithmic on the ViT-Tiny shapes); this one is bounded by bytes.
// Stage = [128 + BN rows][64 B]; 16 B chunk c of row r sits at chunk c ^ f(r>>2), f = {0,2,3,1} (chosen so
// that the 16-lane groups of ds_read_b128, which mix lanes of chunk c and c+1, hit 16 distinct slots): the DMA
// writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address (guide rule 21)
// and again on the ds_read_b128 fragment reads, which are then bank-conflict free.
// Rows past M / N are clamped to the last valid row (their products are never stored).
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// RESK ("residual as K"): for C = A W^T + R with no bias / activation / dropout (the data-gradient GEMMs) and a tile that
// spans the whole row (BN == N), R is streamed through the SAME ring as BN/32 extra K stages and accumulated by MFMAs
// against identity fragments built in registers: acc += R * I.  Exact (x * 1.0 and the same final fp32 add as the
// epilogue's `+ residual`), and the epilogue has no loads left -- what made the 192-column tile lose 9-12 us per launch
// to an exposed residual fetch (2 workgroups per CU) now arrives prefetched like any operand stage.
template <int BMT, int BN, int EPI, bool RESK = false>
// Variants whose tail loads a residual / gate tile issue those loads EARLY (below) and state 2 waves per SIMD to the
// register allocator, which keeps the accumulators in VGPRs: with the default bound the compiler split them over VGPRs
// and AGPRs and, once the last two stages were peeled, shuttled them through v_accvgpr moves inside the K loop.
// Bias-only variants keep the plain loop and the default allocation (99 VGPR + 64 AGPR, three workgroups per CU): for
// them the early form measured SLOWER (FFN1 shape 33.2 -> 38.5 us, QKV 27.7 -> 31.5), with nothing worth prefetching.
__global__ __launch_bounds__(GEMM_THREADS, RESK ? 3 : ((EPI & ~EPI_PE) == 0 ? 1 : 2)) void gemm_nt_async_kernel(const GemmParams p) {
  constexpr int BK2 = 32;
  constexpr int WN = BN / 2, NT = WN / 16, MT = BMT / 32;   // wave tile = BMT/2 rows x BN/2 cols
  constexpr int STAGE_BYTES = (BMT + BN) * BK2 * 2;
  constexpr int NS = 3;
  constexpr int A_LD = BMT * BK2 * 2 / (4 * 1024);     // 1 KiB DMA pieces per wave per stage: A 2 | 1
  constexpr int B_LD = BN * BK2 * 2 / (4 * 1024);      //                                      B 2 | 1
  constexpr int PER_STAGE = A_LD + B_LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int m0 = (t / p.tiles_n) * BMT, n0 = (t % p.tiles_n) * BN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane source rows for this wave's DMA pieces (16 rows x 64 B per piece)
  const int prow = lane >> 2, pch = lane & 3;
  const bf16* a_src[A_LD];
  const bf16* b_src[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int row = (wave * A_LD + i) * 16 + prow;
    const int gm = min(m0 + row, p.M - 1);
    a_src[i] = p.A + (long)gm * p.lda + (pch ^ swz64(row)) * 8;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int row = (wave * B_LD + i) * 16 + prow;
    const int gn = min(n0 + row, p.N - 1);
    b_src[i] = p.B + (long)gn * p.ldb + (pch ^ swz64(row)) * 8;
  }
  const int nk = p.K / BK2;
  const int ntot = nk + (RESK ? BN / BK2 : 0);
  const bf16* r_src[A_LD];
  bf16x8 idf[2];
  if (RESK) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int row = (wave * A_LD + i) * 16 + prow;
      const int gm = min(m0 + row, p.M - 1);
      r_src[i] = p.residual + (long)gm * p.ldr + (pch ^ swz64(row)) * 8;
    }
    // identity fragments (the weight-side MFMA operand): lane (n = lane & 15, k = 8 (lane >> 4) + e) of the 16-column
    // tile h of a 32-wide residual stage holds 1 where 16 h + n == k
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int estar = 16 * h + (lane & 15) - 8 * (lane >> 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) idf[h][e] = (bf16)(e == estar ? 1.0f : 0.0f);
    }
  }
  auto issue = [&](int ks) {
    unsigned char* st = smem + (ks % NS) * STAGE_BYTES;
    if (RESK && ks >= nk) {             // a residual stage: rows of R in the A half, nothing in the B half
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(r_src[i] + (ks - nk) * BK2), (lds_void_t*)(st + (wave * A_LD + i) * 1024), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(a_src[i] + ks * BK2), (lds_void_t*)(st + (wave * A_LD + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(b_src[i] + ks * BK2),
                                       (lds_void_t*)(st + BMT * BK2 * 2 + (wave * B_LD + i) * 1024), 16, 0, 0);
  };

#ifdef IQ_GEMM_STAMPS
#define IQ_STAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.stamps[(long)blockIdx.x * 6 + (i)] = t_; } } while (0)
#else
#define IQ_STAMP(i) do {} while (0)
#endif
  if (p.stagger > 0) {
    // Co-resident workgroups otherwise run their phases in lockstep (all waiting on HBM, then all on the MFMA
    // pipe, then all storing): de-phase them once at launch; the offset persists as slots are refilled.
    const int d = (int)(((unsigned)blockIdx.x * 2654435761u) >> 30) * p.stagger;   // 0..3 x stagger
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8);
  }
  // the tail's registers; the device-resident dropout step is loaded FIRST (oldest entry of the in-order vmcnt queue)
  constexpr int EPIX = RESK ? (EPI & ~EPI_RES) : EPI;
  EpiRegs<MT, NT, EPIX> R;
  R.rng = p.drop_on ? rng_resolve(p.rng) : p.rng;
  IQ_STAMP(0);
  issue(0);
  if (ntot > 1) issue(1);
  // head of every stage: this wave's pieces of stage ks have landed (the youngest stage stays in flight), everyone's
  // have (barrier, which also says stage ks-1 is no longer read), the slot ks-1 vacated is refilled
  auto stage_begin = [&](int ks) {
    if (ks + 1 < ntot) {
      if (!RESK || ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + 2 < ntot) issue(ks + 2);
  };
  const int ch = lane >> 4;
  auto compute = [&](int ks) {
    const bf16* As = reinterpret_cast<const bf16*>(smem + (ks % NS) * STAGE_BYTES);
    const bf16* Bs = As + BMT * BK2;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * (BMT / 2) + i * 16 + (lane & 15);
      af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK2 + (ch ^ swz64(row)) * 8);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WN + j * 16 + (lane & 15);
      bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK2 + (ch ^ swz64(row)) * 8);
    }
#ifdef IQ_NT_NO_MFMA   // ablation build (scripts/dbg/ablate.py): timing only
#pragma unroll
    for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(bfr[j]));
#else
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: see gemm_epilogue
#endif
  };
  const int row0 = m0 + (wave >> 1) * (BMT / 2), col0 = n0 + (wave & 1) * (BN / 2);
  constexpr bool EARLY_TAIL = !RESK && (EPIX & (EPI_RES | EPI_GATE)) != 0;
  if constexpr (EARLY_TAIL) {
    // The tail's global loads (bias, residual, gate) go out as soon as the LAST operand stage has been issued and
    // travel under the last two stages' MFMAs: they are the youngest entries of the in-order vmcnt queue, so the
    // remaining stage waits leave exactly EARLY of them outstanding.  (Issued BEFORE the K loop they made every stage
    // wait behind them -- measured slower, profiles/r01_probes.txt; after the loop their latency was exposed.)
    // The two last stages are peeled: with the register loads inside the loop body the compiler drains the whole
    // queue (vmcnt(0)) on every iteration.
    constexpr int EARLY = epi_early_loads<MT, NT, EPIX>();
    for (int ks = 0; ks + 2 < nk; ++ks) {
      stage_begin(ks);
      if (ks == 0) IQ_STAMP(1);
      compute(ks);
    }
    if (nk >= 2) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE) : "memory");      // stage nk-2 landed, nk-1 in flight
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      epi_load_early<MT, NT, EPIX>(p, R, row0, col0, lane);
      asm volatile("" ::: "memory");                                          // ... issued now, not after this stage's MFMAs
      compute(nk - 2);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(EARLY) : "memory");          // stage nk-1 landed; only the tail's loads fly
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      compute(nk - 1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      epi_load_early<MT, NT, EPIX>(p, R, row0, col0, lane);
      asm volatile("" ::: "memory");
      compute(0);
    }
  } else {
    for (int ks = 0; ks < nk; ++ks) {
      stage_begin(ks);
      if (ks == 0) IQ_STAMP(1);
      compute(ks);
    }
  }
  if (RESK) {
    // residual stage t covers columns [32 t, 32 t + 32) = column tiles 2t and 2t+1 of the row: the wave that owns them
    // (wn == 2t / NT) accumulates R * I into exactly those two tiles.  Fully unrolled: every accumulator index is static
    // (one loop with a run-time tile index made the compiler shuttle all accumulators through copies every K step).
    constexpr int NRS = BN / BK2;
#pragma unroll
    for (int t = 0; t < NRS; ++t) {
      stage_begin(nk + t);
      const bf16* As = reinterpret_cast<const bf16*>(smem + ((nk + t) % NS) * STAGE_BYTES);
      if (wn == (2 * t) / NT) {
        bf16x8 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = wm * (BMT / 2) + i * 16 + (lane & 15);
          af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK2 + (ch ^ swz64(row)) * 8);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int jbase = (2 * t) % NT;      // compile-time after unrolling
#pragma unroll
          for (int i = 0; i < MT; ++i)
            acc[i][jbase + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idf[h], af[i], acc[i][jbase + h], 0, 0, 0);
        }
      }
    }
  }
  IQ_STAMP(2);
  if constexpr (!EARLY_TAIL) {    // bias-only tails (and RESK, whose residual came through the ring): load after the loop
    const IqRng keep = R.rng;
    epi_load<MT, NT, EPIX>(p, R, row0, col0, lane);
    R.rng = keep;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the tail's loads have landed
  IQ_STAMP(3);
  epi_finish<MT, NT, EPIX>(p, acc, R, row0, col0, lane);
  IQ_STAMP(4);
#ifdef IQ_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // only to time the store drain: a wave may retire with stores in flight
#endif
  IQ_STAMP(5);
}

}  // namespace

// all-zero bias for calls without one: the early tail loads are unconditional (gemm_common.h, epi_load_early)
constexpr int ZERO_BIAS_FLOATS = 16384;
__device__ float g_zero_bias[ZERO_BIAS_FLOATS];
static const float* zero_bias() {
  static const float* ptr = [] {
    void* q = nullptr;
    return hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_bias)) == hipSuccess ? (const float*)q : (const float*)nullptr;
  }();
  return ptr;
}

#ifdef IQ_GEMM_STAMPS
static unsigned long long* g_stamps = nullptr;
extern "C" void iq_debug_set_stamps(unsigned long long* p) { g_stamps = p; }
#endif
extern "C" int iq_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                               const iq_epilogue_t* epi, iq_stream_t stream) {
  if (M <= 0 || N <= 0) return IQ_OK;
  if (!A || !B || !C || K <= 0) return IQ_ERR_ARG;
  if ((K % 8) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return IQ_ERR_UNSUPPORTED;
  GemmParams p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = (bf16*)C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
#ifdef IQ_GEMM_STAMPS
  p.stamps = g_stamps;
#endif
  if (epi) {
    p.bias = epi->bias; p.relu = epi->relu;
    p.pe = epi->pe; p.tok = epi->tok; p.seq = epi->seq; p.cls_off = epi->cls_off;
    if (epi->tok < 0 || (epi->tok > 0 && epi->seq < epi->tok + epi->cls_off)) return IQ_ERR_ARG;
    if (epi->drop.p > 0.f) {
      if (epi->drop.p >= 1.f) return IQ_ERR_ARG;
      p.drop_on = 1;
      p.rng.seed = epi->drop.seed; p.rng.step = epi->drop.step; p.rng.site = epi->drop.site;
      p.rng.step_dev = epi->drop.step_dev;
      p.thresh = dropout_thresh(epi->drop.p);
      p.dscale = dropout_scale(epi->drop.p);
    }
    p.gate = (const bf16*)epi->gate; p.ldg = epi->ldg; p.gate_scale = epi->gate_scale;
    p.residual = (const bf16*)epi->residual; p.ldr = epi->ldr;
    if ((p.gate && (p.ldg % 8)) || (p.residual && (p.ldr % 8))) return IQ_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  const bool async_ok = (K % 32 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
  // Column tile: 128 for wide N, else 64 (the register-staged fallback for K % 32 != 0 uses the same choice).
  const bool wide = (N % 128 == 0 || N > 512);
  const int bn = wide ? 128 : 64;
  // Row tile: 128, or 64 when the launch is only a round or two of 128-row tiles (three of them fit a CU: 768 slots): the
  // last, partly filled round then costs a whole tile time.  cfg C (M = 16,640): FFN1 / gate data gradient = 1,040 tiles of
  // 128 x 128 = 1.35 rounds, the QKV projection 390, the N = 128 data gradient 130.  The 64-row tile streams A once as well
  // (the weight is re-read from L2).  cfg B's launches are 1,182..2,364 tiles and stay on 128 rows (64-row tiles measured
  // slower there: more weight re-reads per byte of A).  IQ_TUNE_NT_ROWS = 64 | 128 forces the choice (probes).
  static const int tune_rows = [] { const char* e = getenv("IQ_TUNE_NT_ROWS"); return e ? atoi(e) : 0; }();
  static const int tune_tiles = [] { const char* e = getenv("IQ_TUNE_NT_TILES"); return e ? atoi(e) : 512; }();
  int bm = BM;
  if (async_ok) {
    const long t128 = (long)((M + 127) / 128) * ((N + bn - 1) / bn);
    if (tune_rows == 64 || (tune_rows == 0 && t128 < tune_tiles)) bm = 64;
  }
  p.tiles_m = (M + bm - 1) / bm;
  p.tiles_n = (N + bn - 1) / bn;
  const int grid = p.tiles_m * p.tiles_n;
  int epi_mode = (p.residual ? EPI_RES : 0) | (p.gate ? EPI_GATE : 0) | ((p.pe || p.tok > 0) ? EPI_PE : 0);
  if ((epi_mode & EPI_PE) && (!p.pe || p.tok <= 0 || (epi_mode & (EPI_RES | EPI_GATE)))) return IQ_ERR_UNSUPPORTED;
  if (p.bias && ((uintptr_t)p.bias % 16)) return IQ_ERR_ARG;
  const bool has_bias = p.bias != nullptr;
  if (!has_bias) {
    if (N > ZERO_BIAS_FLOATS || !zero_bias()) return IQ_ERR_UNSUPPORTED;
    p.bias = zero_bias();
  }
  p.stagger = 2;      // measured best of {0, 2, 6} (profiles/r01_probes.txt)
  // algorithmic work of this launch (operands read once, result written once; residual / gate tiles read once)
  const double mn = (double)M * N;
  const double work_bytes = 2.0 * ((double)M * K + (double)N * K + mn) + (p.residual ? 2.0 * mn : 0.0) + (p.gate ? 2.0 * mn : 0.0) +
                            (has_bias ? 4.0 * N : 0.0);
  const double work_flops = 2.0 * mn * K;
  if (async_ok && gemm_big_try(p, epi_mode, st)) {
    IQ_PROF_K(work_bytes, work_flops, "gemm_big_kernel<%d>", epi_mode);
    return iq_launch_status();
  }
  // C = A W^T + R, nothing else in the tail, whole rows in one 192- / 128-column tile: residual streamed as extra K stages.
  // (the plain 192-column tile reads A once -- N=192 K=768: 27.6 vs 37.3 us -- but lost it all to an exposed residual fetch)
  if (async_ok && epi_mode == EPI_RES && (N == 192 || N == 128) && K >= 384 && !has_bias && !p.relu && !p.drop_on &&
      ((uintptr_t)p.residual % 16) == 0) {
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = 1;
    const size_t lds = (size_t)3 * (BM + N) * 32 * 2;       // <= 60 KiB: under the 64 KiB default cap, no attribute call
    if (N == 192) gemm_nt_async_kernel<128, 192, EPI_RES, true><<<p.tiles_m, GEMM_THREADS, lds, st>>>(p);
    else gemm_nt_async_kernel<128, 128, EPI_RES, true><<<p.tiles_m, GEMM_THREADS, lds, st>>>(p);
    IQ_PROF_K(work_bytes, work_flops, "gemm_nt_async_kernel<128, %d, 1, true>", N);
    return iq_launch_status();
  }
  // Variants that were measured and not kept (weight-stationary persistent workgroups, wave-private weight-in-registers
  // waves, the A-stationary persistent row sweep with counted waits across tile boundaries, 64- / 256-row tiles, 192-column
  // tiles with epilogue loads) live in scripts/dbg/variants/ with their numbers.
  const size_t lds_async = (size_t)3 * (bm + bn) * 32 * 2;     // ring of 3 stages
  const size_t lds_reg = (size_t)(BM + bn) * BK * 2;
#define IQ_GEMM_LAUNCH(BN_, EPI_)                                                                  \
  do {                                                                                             \
    if (async_ok && bm == 64) gemm_nt_async_kernel<64, BN_, EPI_><<<grid, GEMM_THREADS, lds_async, st>>>(p);  \
    else if (async_ok) gemm_nt_async_kernel<128, BN_, EPI_><<<grid, GEMM_THREADS, lds_async, st>>>(p);  \
    else gemm_nt_kernel<BN_, EPI_><<<grid, GEMM_THREADS, lds_reg, st>>>(p);                        \
  } while (0)
#define IQ_GEMM_EPI(BN_)                                                       \
  switch (epi_mode) {                                                          \
    case 0: IQ_GEMM_LAUNCH(BN_, 0); break;                                     \
    case EPI_RES: IQ_GEMM_LAUNCH(BN_, EPI_RES); break;                         \
    case EPI_GATE: IQ_GEMM_LAUNCH(BN_, EPI_GATE); break;                       \
    case EPI_RES | EPI_GATE: IQ_GEMM_LAUNCH(BN_, EPI_RES | EPI_GATE); break;   \
    case EPI_PE: IQ_GEMM_LAUNCH(BN_, EPI_PE); break;                           \
    default: return IQ_ERR_UNSUPPORTED;                                        \
  }
  if (bn == 128) { IQ_GEMM_EPI(128) } else { IQ_GEMM_EPI(64) }
#undef IQ_GEMM_EPI
#undef IQ_GEMM_LAUNCH
  if (async_ok) IQ_PROF_K(work_bytes, work_flops, "gemm_nt_async_kernel<%d, %d, %d, false>", bm, bn, epi_mode);
  else IQ_PROF_K(work_bytes, work_flops, "gemm_nt_kernel<%d, %d>", bn, epi_mode);
  return iq_launch_status();
}
