// Two chained bf16 GEMMs in one kernel:   H[M,F] = epi1(X[M,D] * Wa[F,D]^T);   Y[M,D] = epi2(H[M,F] * Wb[D,F]^T)
//
// This is PositionwiseFeedForward.forward (V/models/layers/position_wise_feed_forward.py:12-17, linear1 -> ReLU ->
// dropout -> linear2) followed by the encoder layer's dropout2 + residual (V/models/blocks/encoder_layer.py:30-33), and
// with the transposed weight shadows the same kernel is its data-gradient chain (dY -> gate by hid > 0 -> dX + residual):
// the four widest of the eight GEMMs of an encoder layer's forward + backward.
//
// Why fuse.  The ablations of the tiled kernel (profiles/r01_probes.txt) put the second GEMM's time in streaming its
// K = F = 768 wide A operand through L2 -> LDS (27.5 of 34 us with MFMAs and epilogue compiled out) and ~6 us per launch
// in pipeline fill / drain.  Here H is produced 128 columns at a time, rounded and written to HBM once (the backward
// pass needs it) and consumed straight from LDS as the A operand of the second product, whose accumulators
// (64 x D fp32 per workgroup) stay in registers across the whole hidden dimension: H is never read back, X is read
// once, one launch instead of two.
//
// Workgroup = 64 rows, 4 waves (2 x 2).  Step 1 per hidden chunk: wave tile 32 x 64 (acc 32 VGPRs), X fragments from a
// resident LDS image, Wa k-stages through the DMA ring.  Step 2: wave tile 32 x D/2 (acc D/4 VGPRs), H fragments from
// the LDS image the first epilogue left, Wb k-stages through the same ring.  The weight stream (Wa and Wb chunk by
// chunk, 10 stages per chunk for D = 192) is one continuous 3-slot global_load_lds ring, L2-resident (590 KB per
// layer), prefetched across chunk boundaries.  Both epilogues are the register-only gemm_epilogue of gemm_common.h.
//
// MEASURED (cfg B, M = 50432, D = 192, F = 768): forward+dropout 75.6 us vs 74.3 us for the two launches, eval 66.6 vs
// 68.0, data-gradient chain 71.7 vs 78.9; inside the training step no gain.  What it saves in activation traffic (H is
// not read back: 77 MB) it pays in weight traffic: every 64-row block re-streams the layer's 590 KB of weights from
// L2 (454 MB per call), two stages in flight per workgroup, so the stream is latency-bound.  The next version needs
// 128-row blocks with 8 waves and a 5-slot ring (half the weight bytes, twice the bytes in flight).  Not used by the
// model plan by default (IQ_BWD_CHAIN=1 routes the FFN data-gradient pair through it).
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int CH_THREADS = 256, CH_BM = 64, CH_BN1 = 128, CH_BK = 32, CH_NS = 3;
constexpr int CH_HS_BYTES = CH_BM * CH_BN1 * 2;       // 16 KiB: H chunk image, 256 B rows

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int ch_swz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}

template <int D, int EPI1, int EPI2>
__global__ __launch_bounds__(CH_THREADS, 2) void gemm_chain_kernel(const GemmParams p1, const GemmParams p2) {
  constexpr int KSA = D / CH_BK;            // k-stages of step 1 (contraction over D)
  constexpr int KSB = CH_BN1 / CH_BK;       // k-stages of step 2 per hidden chunk (4)
  constexpr int PA = CH_BN1 * CH_BK * 2 / 1024 / 4;   // DMA instructions per wave per Wa stage (2)
  constexpr int PB = D * CH_BK * 2 / 1024 / 4;        // ... per Wb stage (D = 192: 3)
  constexpr int STAGE_BYTES = (D > CH_BN1 ? D : CH_BN1) * CH_BK * 2;
  constexpr int XROW = D * 2;               // bytes per row of the X image
  constexpr int NT2 = D / 32;               // 16-col tiles of the step-2 wave tile (32 x D/2)
  static_assert(D % 64 == 0 && D <= 256, "d_model must be 64, 128, 192 or 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;                                 // [64][D] bf16, chunk c of row r at c ^ ((r >> 1) & 7)
  unsigned char* Hs = Xs + CH_BM * XROW;                    // [64][128] bf16, chunk c of row r at c ^ (r & 15)
  unsigned char* ring = Hs + CH_HS_BYTES;                   // 3 stages of [rows][64 B], chunk ^ swz64(row)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * CH_BM;
  const int F = p1.N, nchunk = F / CH_BN1;
  const int Q = nchunk * (KSA + KSB);      // weight stages this workgroup consumes

  // ---- X tile -> LDS (once) -----------------------------------------------------------------------------------
  {
    constexpr int NI = CH_BM * XROW / 1024;             // 1 KiB DMA pieces (D = 192: 24)
#pragma unroll
    for (int i = wave; i < NI; i += 4) {
      const int off = i * 1024 + lane * 16;
      const int r = off / XROW, pc = (off - r * XROW) >> 4;
      const int lc = pc ^ ((r >> 1) & 7);
      const int gm = min(m0 + r, p1.M - 1);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p1.A + (long)gm * p1.lda + lc * 8), (lds_void_t*)(Xs + i * 1024), 16, 0, 0);
    }
  }
  // ---- weight stage q: chunk c = q / (KSA+KSB); s < KSA: Wa rows [c*128, +128), k = 32 s; else Wb rows [0, D), k = c*128 + 32 (s-KSA)
  const int prow = lane >> 2, pch = lane & 3;
  auto issue = [&](int q) {
    const int c = q / (KSA + KSB), s = q - c * (KSA + KSB);
    unsigned char* dst = ring + (q % CH_NS) * STAGE_BYTES;
    if (s < KSA) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = (wave * PA + i) * 16 + prow;
        const bf16* src = p1.B + (long)(c * CH_BN1 + r) * p1.ldb + s * CH_BK + ((pch ^ ch_swz64(r)) << 3);
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(dst + (wave * PA + i) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int r = (wave * PB + i) * 16 + prow;
        const bf16* src = p2.B + (long)r * p2.ldb + c * CH_BN1 + (s - KSA) * CH_BK + ((pch ^ ch_swz64(r)) << 3);
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(dst + (wave * PB + i) * 1024), 16, 0, 0);
      }
    }
  };
  issue(0);
  issue(1);

  f32x4 acc2[2][NT2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ch = lane >> 4, r16 = lane & 15;
  int q = 0;
  for (int c = 0; c < nchunk; ++c) {
    // ================= step 1: H chunk = X * Wa[c]^T =================
    f32x4 acc1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSA; ++s, ++q) {
      // stage q landed for this wave's pieces (the younger stage stays in flight), then for everyone.  The very first
      // wait also covers the X tile (older).  Stores of an epilogue are older than any stage waited for here.
      if (s + 1 < KSA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (q + 2 < Q) issue(q + 2);
      const bf16* Ws = reinterpret_cast<const bf16*>(ring + (q % CH_NS) * STAGE_BYTES);
      bf16x8 xf[2], wf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + r16;
        const int lc = (s * 4 + ch) ^ ((row >> 1) & 7);
        xf[i] = *reinterpret_cast<const bf16x8*>(Xs + row * XROW + (lc << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + r16;
        wf[j] = *reinterpret_cast<const bf16x8*>(Ws + row * CH_BK + ((ch ^ ch_swz64(row)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc1[i][j], 0, 0, 0);
    }
    // everyone has read the previous chunk's H image (step 2 of chunk c-1 ended before this chunk's first barrier)
    gemm_epilogue<2, 4, EPI1, true>(p1, acc1, m0 + wm * 32, c * CH_BN1 + wn * 64, lane, Hs, wm * 32, wn * 64);
    // ================= step 2: Y += H chunk * Wb[:, c]^T =================
#pragma unroll
    for (int s = 0; s < KSB; ++s, ++q) {
      // The epilogue above retired every older DMA (its own vmcnt(0)) before storing: stages b0 and b1 are in LDS
      // already; b2 / b3 were requested after those stores.
      if (s >= 2) {
        if (q + 1 < Q) {
          if (s + 1 < KSB) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __builtin_amdgcn_s_barrier();       // s == 0: also publishes the H image
      asm volatile("" ::: "memory");
      if (q + 2 < Q) issue(q + 2);
      const bf16* Ws = reinterpret_cast<const bf16*>(ring + (q % CH_NS) * STAGE_BYTES);
      bf16x8 hf[2], wf[NT2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + r16;
        const int lc = (s * 4 + ch) ^ (row & 15);
        hf[i] = *reinterpret_cast<const bf16x8*>(Hs + row * 256 + (lc << 4));
      }
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const int row = wn * (D / 2) + j * 16 + r16;
        wf[j] = *reinterpret_cast<const bf16x8*>(Ws + row * CH_BK + ((ch ^ ch_swz64(row)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], hf[i], acc2[i][j], 0, 0, 0);
    }
  }
  gemm_epilogue<2, NT2, EPI2>(p2, acc2, m0 + wm * 32, wn * (D / 2), lane);
}

}  // namespace

// Fill a GemmParams epilogue part from the ABI struct (same rules as iq_gemm_bf16_nt).
static int fill_epi(GemmParams& p, const iq_epilogue_t* epi) {
  if (!epi) return IQ_OK;
  if (epi->pe || epi->tok) return IQ_ERR_UNSUPPORTED;
  p.bias = epi->bias; p.relu = epi->relu;
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
  if (p.bias && ((uintptr_t)p.bias % 16)) return IQ_ERR_ARG;
  return IQ_OK;
}

extern "C" int iq_gemm_chain_supported(int D, int F) {
  return (D == 64 || D == 128 || D == 192 || D == 256) && F > 0 && (F % CH_BN1) == 0;
}

extern "C" int iq_gemm_bf16_chain(const void* X, int ldx, const void* Wa, int ldwa, void* H, int ldh, const void* Wb,
                                  int ldwb, void* Y, int ldy, int M, int F, int D, const iq_epilogue_t* epi1,
                                  const iq_epilogue_t* epi2, iq_stream_t stream) {
  if (M <= 0) return IQ_OK;
  if (!X || !Wa || !H || !Wb || !Y) return IQ_ERR_ARG;
  if (!iq_gemm_chain_supported(D, F)) return IQ_ERR_UNSUPPORTED;
  if ((ldx % 8) || (ldwa % 8) || (ldh % 8) || (ldwb % 8) || (ldy % 8)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)X | (uintptr_t)Wa | (uintptr_t)Wb | (uintptr_t)H | (uintptr_t)Y) % 16) return IQ_ERR_ARG;
  GemmParams p1 = {}, p2 = {};
  p1.A = (const bf16*)X; p1.lda = ldx; p1.B = (const bf16*)Wa; p1.ldb = ldwa; p1.C = (bf16*)H; p1.ldc = ldh;
  p1.M = M; p1.N = F; p1.K = D;
  p2.A = (const bf16*)H; p2.lda = ldh; p2.B = (const bf16*)Wb; p2.ldb = ldwb; p2.C = (bf16*)Y; p2.ldc = ldy;
  p2.M = M; p2.N = D; p2.K = F;
  int rc = fill_epi(p1, epi1);
  if (rc != IQ_OK) return rc;
  rc = fill_epi(p2, epi2);
  if (rc != IQ_OK) return rc;
  if (p1.residual || p2.gate) return IQ_ERR_UNSUPPORTED;     // step 1: bias / ReLU / dropout / gate; step 2: bias / dropout / residual
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  const int grid = (M + CH_BM - 1) / CH_BM;
  const int e1 = p1.gate ? EPI_GATE : 0, e2 = p2.residual ? EPI_RES : 0;
#define IQ_CH_LAUNCH(D_, E1_, E2_)                                                                                    \
  do {                                                                                                                \
    constexpr size_t lds = (size_t)CH_BM * D_ * 2 + CH_HS_BYTES + (size_t)CH_NS * (D_ > CH_BN1 ? D_ : CH_BN1) * CH_BK * 2; \
    auto k = gemm_chain_kernel<D_, E1_, E2_>;                                                                         \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
    k<<<grid, CH_THREADS, lds, st>>>(p1, p2);                                                                         \
  } while (0)
#define IQ_CH_EPI(D_)                                                         \
  do {                                                                        \
    if (e1 == 0 && e2 == 0) IQ_CH_LAUNCH(D_, 0, 0);                           \
    else if (e1 == 0) IQ_CH_LAUNCH(D_, 0, EPI_RES);                           \
    else if (e2 == 0) IQ_CH_LAUNCH(D_, EPI_GATE, 0);                          \
    else IQ_CH_LAUNCH(D_, EPI_GATE, EPI_RES);                                 \
  } while (0)
  switch (D) {
    case 64: IQ_CH_EPI(64); break;
    case 128: IQ_CH_EPI(128); break;
    case 192: IQ_CH_EPI(192); break;
    default: IQ_CH_EPI(256); break;
  }
#undef IQ_CH_EPI
#undef IQ_CH_LAUNCH
  return iq_launch_status();
}
