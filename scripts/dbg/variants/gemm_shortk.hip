// bf16 MFMA GEMM for short contractions, C[M,N] = epilogue(A[M,K] * B[N,K]^T) with K = 128 | 192 (gfx950): the packed QKV
// projection, the first FFN GEMM and the FFN2 data gradient of ViT-Tiny and of the raw-IQ encoder
// (multi_head_attention.py:18, position_wise_feed_forward.py:13; 36 launches of a cfg B step).
//
// What the ring kernel (gemm_nt.hip) pays on these shapes is the operand stream through L2 -> LDS, and the LDS-DMA path
// moves whole 128-byte lines 1.6x faster than the half lines a 32-deep stage asks for (scripts/dbg/dma_probe: 8 rows x
// 128 B per wave-instruction 42 B/clk/CU from L2, 16 rows x 64 B 27 B/clk/CU; the second half of every line is fetched
// again two stages later).  With K <= 192 there is no need for a ring at all:
//   * operand units of 128 rows x 64 k (128-byte rows, 16 KiB, 16 DMA pieces of 8 rows x 128 B), A0 B0 A1 B1 A2 B2;
//   * five slots (80 KiB, two workgroups per CU): the first five units are requested before anything else, the sixth
//     (B2) goes into B0's slot as soon as the first 64-deep step has been multiplied -- 160 KiB in flight per CU
//     instead of 96, every line fetched once;
//   * rows XOR-swizzled on the global side (chunk ^ (row >> 1) & 7): conflict-free ds_read_b128 on 128-byte rows;
//   * the tail's loads (bias, residual, gate) go out behind the last unit and travel under the remaining MFMAs
//     (counted vmcnt), the register-only epilogue of gemm_common.h follows.
// 128 x 128 tile, 4 waves as 2 x 2, 64 x 64 per wave, 32 MFMAs per wave per 64-deep step.
#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int SK_THREADS = 256, SK_BM = 128, SK_BN = 128, SK_UNIT = 128 * 128;   // bytes per operand unit
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
constexpr int sk_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int EPI, int NK>    // NK = K / 64: 2 | 3
__global__ __launch_bounds__(SK_THREADS, 2) void gemm_shortk_kernel(const GemmParams p) {
  constexpr int MT = 4, NT = 4, PU = 4;                 // DMA pieces per wave per unit
  constexpr int E = epi_early_loads<MT, NT, EPI>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int m0 = (t / p.tiles_n) * SK_BM, n0 = (t % p.tiles_n) * SK_BN;

  if (p.stagger > 0) {          // de-phase co-resident workgroups once (gemm_nt.hip)
    const int d = (int)(((unsigned)blockIdx.x * 2654435761u) >> 30) * p.stagger;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8);
  }
  EpiRegs<MT, NT, EPI> R;
  R.rng = p.drop_on ? rng_resolve(p.rng) : p.rng;       // oldest entry of the vector-memory queue

  // this wave's pieces of a unit: pieces 4w..4w+3 = rows 32w..32w+31, 8 rows x 128 B each; lane -> (row, 16 B chunk)
  const int prow = lane >> 3, pch = lane & 7;
  const bf16* a_src[PU];
  const bf16* b_src[PU];
#pragma unroll
  for (int i = 0; i < PU; ++i) {
    const int row = (wave * PU + i) * 8 + prow;
    const int sw = (pch ^ ((row >> 1) & 7)) * 8;
    a_src[i] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + sw;
    b_src[i] = p.B + (long)min(n0 + row, p.N - 1) * p.ldb + sw;
  }
  auto issue_unit = [&](const bf16* const (&src)[PU], int kt, int slot) {
#pragma unroll
    for (int i = 0; i < PU; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(src[i] + kt * 64), (lds_void_t*)(smem + slot * SK_UNIT + (wave * PU + i) * 1024),
                                       16, 0, 0);
  };
  // slots: A_kt -> 2 kt, B_kt -> 2 kt + 1, except B2 -> 1 (B0's, refilled after the first step)
  issue_unit(a_src, 0, 0);
  issue_unit(b_src, 0, 1);
  issue_unit(a_src, 1, 2);
  issue_unit(b_src, 1, 3);
  if (NK == 3) issue_unit(a_src, 2, 4);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ch = lane >> 4, c16 = lane & 15;
  auto compute = [&](int aslot, int bslot) {
    const unsigned char* As = smem + aslot * SK_UNIT;
    const unsigned char* Bs = smem + bslot * SK_UNIT;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * 64 + i * 16 + c16;
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * 128 + (((h * 4 + ch) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * 64 + j * 16 + c16;
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + (((h * 4 + ch) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: gemm_common.h
    }
  };
  const int row0 = m0 + wm * 64, col0 = n0 + wn * 64;

  // ---- step 0: A0, B0 landed; 2 (NK = 2) or 3 units stay in flight -------------------------------------------------
  __builtin_amdgcn_s_waitcnt(sk_vmcnt((NK == 3 ? 3 : 2) * PU));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  compute(0, 1);
  if (NK == 3) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's reads of B0 have returned, not merely been issued)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                       // every wave has read B0: its slot takes B2
    asm volatile("" ::: "memory");
    issue_unit(b_src, 2, 1);
  }
  // the tail's loads: youngest entries of the queue from here on
  epi_load_early<MT, NT, EPI>(p, R, row0, col0, lane);
  asm volatile("" ::: "memory");
  // ---- step 1: A1, B1 landed; younger: (A2, B2,) the tail's loads ---------------------------------------------------
  __builtin_amdgcn_s_waitcnt(sk_vmcnt((NK == 3 ? 2 * PU : 0) + E));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  compute(2, 3);
  if (NK == 3) {
    __builtin_amdgcn_s_waitcnt(sk_vmcnt(E));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(4, 1);
  }
  __builtin_amdgcn_s_waitcnt(sk_vmcnt(0));
  epi_finish<MT, NT, EPI>(p, acc, R, row0, col0, lane);
}

}  // namespace

// Called by iq_gemm_bf16_nt with its resolved parameters (bias non-null).  Returns false when the shape / epilogue is not
// this kernel's; true after a launch.
bool gemm_shortk_try(const GemmParams& p0, int epi_mode, hipStream_t st) {
  if (epi_mode != 0 && epi_mode != EPI_RES && epi_mode != EPI_GATE) return false;
  if ((p0.K != 128 && p0.K != 192) || !(p0.N % 128 == 0 || p0.N > 512) || p0.M < 4096) return false;
  if ((((uintptr_t)p0.A | (uintptr_t)p0.B | (uintptr_t)p0.C | (uintptr_t)p0.bias) % 16) || (p0.lda % 8) || (p0.ldb % 8)) return false;
  GemmParams p = p0;
  p.tiles_m = (p.M + SK_BM - 1) / SK_BM;
  p.tiles_n = (p.N + SK_BN - 1) / SK_BN;
  const int grid = p.tiles_m * p.tiles_n;
  const int nk = p.K / 64;
  const size_t lds = (size_t)(nk == 3 ? 5 : 4) * SK_UNIT;
#define IQ_SK_LAUNCH(E_)                                                                                               \
  do {                                                                                                                \
    if (nk == 3) {                                                                                                    \
      auto k = gemm_shortk_kernel<E_, 3>;                                                                             \
      static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * SK_UNIT); \
      (void)attr;                                                                                                     \
      k<<<grid, SK_THREADS, lds, st>>>(p);                                                                            \
    } else {                                                                                                          \
      auto k = gemm_shortk_kernel<E_, 2>;                                                                             \
      static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SK_UNIT); \
      (void)attr;                                                                                                     \
      k<<<grid, SK_THREADS, lds, st>>>(p);                                                                            \
    }                                                                                                                 \
  } while (0)
  if (epi_mode == EPI_RES) IQ_SK_LAUNCH(EPI_RES);
  else if (epi_mode == EPI_GATE) IQ_SK_LAUNCH(EPI_GATE);
  else IQ_SK_LAUNCH(0);
#undef IQ_SK_LAUNCH
  return true;
}
