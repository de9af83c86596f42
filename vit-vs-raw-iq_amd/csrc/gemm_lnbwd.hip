// Data-gradient GEMM with the LayerNorm BACKWARD that consumes it fused into its epilogue (gfx950):
//
//     dX[M,D]  = A[M,K] * Wt[D,K]^T + R                      (fp32, never leaves the registers)
//     g        = dX * gamma ;  xhat = (Z - mean) * rstd
//     dZ       = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat))          bf16 [M,D]
//     dY       = dropout_mask(dZ) * scale                                   bf16 [M,D]  (only when the site drops)
//     partial[tile] = column sums over the tile's rows of (dX * xhat | dX)  fp32 [2*D]  (dgamma | dbeta partials)
//
// This is the autograd backward of `x = norm(dropout(sub_layer(x)) + x)` (V/models/blocks/encoder_layer.py:24-25,32-33,
// LayerNorm: V/models/layers/layers_norm.py:11-19) glued to the GEMM that produces its incoming gradient: the FFN1 data
// gradient feeds norm1's backward, the QKV data gradient feeds norm2's backward of the layer below.  Unfused, that GEMM
// stored dX (M*D*2 B), and iq_ln_bwd re-read it; here the rounded dX tile crosses LDS instead of HBM: 23 launches and
// 2*M*D*2 bytes per LayerNorm disappear.  The gamma / beta partial rows join the layer's slab reduce as before.
//
// Whole-row tile 128 x D (D = 128 | 192), 4 waves 2 x 2, the RESK main loop of gemm_nt.hip: operands AND the residual
// arrive through one global_load_lds ring (residual stages are accumulated against identity fragments).
// Tail: the accumulators are rounded to bf16 (as the unfused GEMM stored them) into a row-major LDS image over the
// drained ring, and the workgroup then runs ln_bwd_kernel's own row-per-lane-group arithmetic on it (LPR lanes x NV
// 16-byte vectors per row, same operation order: dZ / dY are bit-identical to GEMM-then-iq_ln_bwd).  Doing the
// LayerNorm in the MFMA register layout instead (rows spread over two waves, 96 accumulators + 48 Z registers + 96
// recomputed xhat live at once) spilled 26-98 registers whatever was tried.  The Z rows are requested behind the last
// ring stage (counted vmcnt), mean / rstd / gamma wait in LDS from the start of the kernel.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int LB_THREADS = 256;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

struct LnBwdParams {
  const bf16* A; const bf16* B; const bf16* residual;     // [M,K], [D,K], [M,D]
  int lda, ldb, ldr, M, K;
  const bf16* Z; const float* mean; const float* rstd; const float* gamma;
  bf16* dZ; bf16* dY;
  float* partial;                                           // [tiles][2*D]
  int drop_on; IqRng rng; uint32_t thresh; float dscale;
};

__device__ __forceinline__ int bswz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}: gemm_nt.hip

template <int BMT, int BN>
__global__ __launch_bounds__(LB_THREADS, 2) void gemm_lnbwd_kernel(const LnBwdParams p) {
  constexpr int BK2 = 32, NS = 3;
  constexpr int WN = BN / 2, NT = WN / 16, NP = NT / 2, MT = BMT / 32;
  constexpr int STAGE_BYTES = (BMT + BN) * BK2 * 2;
  constexpr int A_LD = BMT * BK2 * 2 / (4 * 1024);
  constexpr int B_LD = BN * BK2 * 2 / (4 * 1024);
  constexpr int PER_STAGE = A_LD + B_LD;
  constexpr int NRS = BN / BK2;                      // residual stages
  constexpr int TAIL_LOADS = (BMT / (4 * (64 / ((BN == 192) ? 8 : 16)))) * (BN / (8 * ((BN == 192) ? 8 : 16)));   // ITER * NV Z vectors
  constexpr int N = BN;
  static_assert(NT % 2 == 0 && NRS >= 2, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const int m0 = blockIdx.x * BMT;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int prow = lane >> 2, pch = lane & 3;
  const bf16* a_src[A_LD];
  const bf16* r_src[A_LD];
  const bf16* b_src[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int row = (wave * A_LD + i) * 16 + prow;
    const int gm = min(m0 + row, p.M - 1);
    a_src[i] = p.A + (long)gm * p.lda + (pch ^ bswz64(row)) * 8;
    r_src[i] = p.residual + (long)gm * p.ldr + (pch ^ bswz64(row)) * 8;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int row = (wave * B_LD + i) * 16 + prow;
    b_src[i] = p.B + (long)row * p.ldb + (pch ^ bswz64(row)) * 8;
  }
  const int nk = p.K / BK2;               // >= 2
  // identity fragments (the weight-side MFMA operand of a residual stage): gemm_nt.hip
  bf16x8 idf[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int estar = 16 * h + (lane & 15) - 8 * (lane >> 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) idf[h][e] = (bf16)(e == estar ? 1.0f : 0.0f);
  }
  auto issue = [&](int ks) {
    unsigned char* st = smem + (ks % NS) * STAGE_BYTES;
    if (ks >= nk) {                        // a residual stage: rows of R in the A half
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

  // the tail works in LayerNorm's layout (layernorm.hip): a row is owned by LPR lanes x NV 16-byte vectors
  constexpr int LPR = (BN == 192) ? 8 : 16, NV = BN / (8 * LPR);
  constexpr int RPW = 64 / LPR, RPB = 4 * RPW, ITER = BMT / RPB;      // rows per wave / per pass, passes per tile
  constexpr int LDI = BN + 8;                                          // padded image row (elements): conflict-free 16 B chunks
  static_assert(BMT * LDI * 2 <= NS * STAGE_BYTES && RPB * 2 * BN * 4 <= NS * STAGE_BYTES, "the tail's LDS fits in the ring");
  const int lj = lane % LPR, rsub = lane / LPR;
  bf16x8 zr[ITER][NV];
  auto issue_tail = [&]() {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int gm = min(m0 + it * RPB + wave * RPW + rsub, p.M - 1);
#pragma unroll
      for (int v = 0; v < NV; ++v) zr[it][v] = *reinterpret_cast<const bf16x8*>(p.Z + (long)gm * N + (v * LPR + lj) * 8);
    }
  };

  const IqRng rng = p.drop_on ? rng_resolve(p.rng) : p.rng;     // oldest entry of the vmcnt queue
  // The tile's small operands -- mean and rstd of its 128 rows, gamma -- go to LDS behind the ring (`side`), loaded
  // before the first stage so that they are the oldest entries of the queue: held in registers until the tail (they were
  // 40 of them) the kernel spilled.
  float* side = reinterpret_cast<float*>(smem + NS * STAGE_BYTES);     // [mean BMT][rstd BMT][gamma N]
  const int sr = tid & (BMT - 1);
  const int sgm = min(m0 + sr, p.M - 1);
  float side_v = 0.f;
  if (tid < 2 * BMT) side_v = tid < BMT ? p.mean[sgm] : p.rstd[sgm];
  f32x4 side_g = {0.f, 0.f, 0.f, 0.f};
  if (tid < N / 4) side_g = *reinterpret_cast<const f32x4*>(p.gamma + tid * 4);
  {
    const int d = (int)(((unsigned)blockIdx.x * 2654435761u) >> 30) * 2;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8);
  }
  issue(0);
  issue(1);
  if (tid < 2 * BMT) side[tid] = side_v;                 // tid < BMT: mean[r] at r; else rstd[r] at BMT + r
  if (tid < N / 4) *reinterpret_cast<f32x4*>(side + 2 * BMT + tid * 4) = side_g;
  const int ch = lane >> 4;
  // main stages: stage ks landed (stage ks+1 stays in flight), barrier, refill the vacated slot
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD) : "memory");            // the next stage is a residual stage
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(ks + 2);                                                                // ks + 2 < ntot always (NRS >= 2)
    const bf16* As = reinterpret_cast<const bf16*>(smem + (ks % NS) * STAGE_BYTES);
    const bf16* Bs = As + BMT * BK2;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * (BMT / 2) + i * 16 + (lane & 15);
      af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK2 + (ch ^ bswz64(row)) * 8);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WN + j * 16 + (lane & 15);
      bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK2 + (ch ^ bswz64(row)) * 8);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
  // residual stages (fully unrolled: static accumulator indices).  Stage t covers columns [32 t, 32 t + 32).
  auto residual_stage = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const bf16* As = reinterpret_cast<const bf16*>(smem + ((nk + t) % NS) * STAGE_BYTES);
    if (wn == (2 * t) / NT) {
      bf16x8 af[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * (BMT / 2) + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK2 + (ch ^ bswz64(row)) * 8);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr int jbase = (2 * t) % NT;
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][jbase + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idf[h], af[i], acc[i][jbase + h], 0, 0, 0);
      }
    }
  };
  auto unrolled = [&](auto self, auto tc) -> void {
    constexpr int t = decltype(tc)::value;
    if constexpr (t < NRS) {
      // stage nk + t landed?  younger entries of the queue: the next residual stage, or the tail's loads, or nothing
      if constexpr (t + 2 < NRS) {               // a later stage will still be issued here
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(nk + t + 2);
      } else if constexpr (t + 2 == NRS) {       // every stage is issued: the tail's loads go behind the last one
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_tail();
        asm volatile("" ::: "memory");
      } else {                                   // last stage: only the tail's loads stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TAIL_LOADS) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      residual_stage(tc);
      self(self, std::integral_constant<int, t + 1>{});
    }
  };
  unrolled(unrolled, std::integral_constant<int, 0>{});

  // ---- tail ------------------------------------------------------------------------------------------------------------
  // (1) dX, rounded to bf16 as the unfused GEMM stored it, into a row-major image over the drained ring
  bf16* img = reinterpret_cast<bf16*>(smem);
  const int row0l = wm * (BMT / 2), col0 = wn * WN;
  __builtin_amdgcn_s_barrier();                  // every wave has left the operand ring
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      float w[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float va = acc[i][2 * jp][r], vb = acc[i][2 * jp + 1][r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        w[r] = __uint_as_float(sw[0]);
        w[4 + r] = __uint_as_float(sw[1]);
      }
      const int col = col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g);
      *reinterpret_cast<bf16x8*>(img + (row0l + i * 16 + c16) * LDI + col) = pack8(w);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // (2) ln_bwd_kernel's arithmetic, dX from the image, Z from the registers requested behind the last stage
  float gmm[NV][8], ag[NV][8], ab[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(side + 2 * BMT + (v * LPR + lj) * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(side + 2 * BMT + (v * LPR + lj) * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gmm[v][e] = a[e]; gmm[v][4 + e] = b[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[v][e] = 0.f; ab[v][e] = 0.f; }
  }
  const float invD = 1.0f / (float)N;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int rl = it * RPB + wave * RPW + rsub;
    const long row = (long)m0 + rl;
    const bool ok = row < p.M;
    const float mean = side[rl], rstd = side[BMT + rl];
    float xh[NV][8], dy[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      bf16x8 td = *reinterpret_cast<const bf16x8*>(img + rl * LDI + (v * LPR + lj) * 8);
      bf16x8 tz = zr[it][v];
      if (!ok) { td = bf16x8{}; tz = bf16x8{}; }
      unpack8(tz, xh[v]);
      unpack8(td, dy[v]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[v][e] = ok ? (xh[v][e] - mean) * rstd : 0.f;
        ag[v][e] += dy[v][e] * xh[v][e];
        ab[v][e] += dy[v][e];
        dy[v][e] *= gmm[v][e];
        s1 += dy[v][e];
        s2 += dy[v][e] * xh[v][e];
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float c1 = s1 * invD, c2 = s2 * invD;
    if (ok) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (dy[v][e] - c1 - xh[v][e] * c2);
        const long off = row * N + (v * LPR + lj) * 8;
        *reinterpret_cast<bf16x8*>(p.dZ + off) = pack8(o);
        if (p.drop_on) {
          const uint32_t keep = dropout_keep8(rng, (uint64_t)off >> 3, p.thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((keep >> e) & 1u) ? o[e] * p.dscale : 0.f;
          *reinterpret_cast<bf16x8*>(p.dY + off) = pack8(o);
        }
      }
    }
  }
  // (3) column sums of the tile: per-lane sums -> LDS (over the image, no longer read) -> one partial row per workgroup
  float* red = reinterpret_cast<float*>(smem);    // [RPB][2*N]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int rslot = wave * RPW + rsub;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + lj) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[rslot * 2 * N + c + e] = ag[v][e];
      red[rslot * 2 * N + N + c + e] = ab[v][e];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int c = tid; c < 2 * N; c += LB_THREADS) {
    float sum = 0.f;
    for (int r = 0; r < RPB; ++r) sum += red[r * 2 * N + c];
    p.partial[(long)blockIdx.x * 2 * N + c] = sum;
  }
}

template <int BMT, int BN>
int launch(const LnBwdParams& p, hipStream_t st) {
  const size_t lds = (size_t)3 * (BMT + BN) * 32 * 2 + (2 * BMT + BN) * sizeof(float);   // ring + mean | rstd | gamma
  auto k = gemm_lnbwd_kernel<BMT, BN>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<<<(p.M + BMT - 1) / BMT, LB_THREADS, lds, st>>>(p);
  return iq_launch_status();
}

// Row block: 128, or 64 where 128-row blocks would leave CUs without a workgroup (cfg C: M = 16,640 = 130 blocks on 256
// CUs).  dZ / dY do not depend on it (row-local arithmetic); the gamma / beta partial rows are one per block, so the caller
// sizes and reduces iq_gemm_lnbwd_partial_rows(M) rows.  IQ_TUNE_LNBWD_ROWS forces it (probes).
inline int lnbwd_block_rows(int M) {
  static const int tune_rows = [] { const char* e = getenv("IQ_TUNE_LNBWD_ROWS"); return e ? atoi(e) : 0; }();
  if (tune_rows == 64 || tune_rows == 128) return tune_rows;
  return (M + 127) / 128 <= 320 ? 64 : 128;
}

}  // namespace

extern "C" int iq_gemm_lnbwd_supported(int D, int K) { return ((D == 128 || D == 192) && K >= 64 && K % 32 == 0) ? 1 : 0; }
extern "C" int iq_gemm_lnbwd_partial_rows(int M) {
  if (M <= 0) return 0;
  const int r = lnbwd_block_rows(M);
  return (M + r - 1) / r;
}

extern "C" int iq_gemm_bf16_lnbwd(const void* A, int lda, const void* Wt, int ldw, const void* residual, int ldr,
                                  const void* z, const float* mean, const float* rstd, const float* gamma,
                                  const iq_dropout_t* drop, void* dz, void* dy, float* partial, int M, int D, int K,
                                  iq_stream_t stream) {
  if (M <= 0) return IQ_OK;
  if (!A || !Wt || !residual || !z || !mean || !rstd || !gamma || !dz || !partial) return IQ_ERR_ARG;
  if (!iq_gemm_lnbwd_supported(D, K) || (lda % 8) || (ldw % 8) || (ldr % 8)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)A | (uintptr_t)Wt | (uintptr_t)residual | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)dy | (uintptr_t)gamma) % 16)
    return IQ_ERR_ARG;
  LnBwdParams p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)Wt; p.residual = (const bf16*)residual;
  p.lda = lda; p.ldb = ldw; p.ldr = ldr; p.M = M; p.K = K;
  p.Z = (const bf16*)z; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
  p.dZ = (bf16*)dz; p.dY = (bf16*)dy; p.partial = partial;
  if (drop && drop->p > 0.f) {
    if (drop->p >= 1.f || !dy) return IQ_ERR_ARG;
    p.drop_on = 1;
    p.rng.seed = drop->seed; p.rng.step = drop->step; p.rng.site = drop->site; p.rng.step_dev = drop->step_dev;
    p.thresh = dropout_thresh(drop->p);
    p.dscale = dropout_scale(drop->p);
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  // A + Wt + residual + Z read, dZ (+ dY) written
  IQ_PROF_K(2.0 * ((double)M * K + (double)D * K + (double)M * D * (3 + (p.drop_on ? 1 : 0))) + 8.0 * M, 2.0 * (double)M * D * K,
            "gemm_lnbwd_kernel<%d, %d>", lnbwd_block_rows(M), D);
  if (lnbwd_block_rows(M) == 64) return D == 192 ? launch<64, 192>(p, st) : launch<64, 128>(p, st);
  return D == 192 ? launch<128, 192>(p, st) : launch<128, 128>(p, st);
}
