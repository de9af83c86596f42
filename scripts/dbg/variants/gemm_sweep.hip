// bf16 MFMA GEMM for wide outputs and short contractions, C[M,N] = epilogue(A[M,K] * B[N,K]^T), N % 128 == 0,
// K <= 192 (gfx950).  The packed QKV projection, the first FFN GEMM and the FFN2 data gradient of ViT-Tiny
// (multi_head_attention.py:18, position_wise_feed_forward.py:13; N = 576 | 768, K = 192, M = frames x tokens).
//
// What limits the tiled kernel (gemm_nt.hip) on these shapes is neither HBM (traffic 1.02 x algorithmic) nor the MFMA
// pipe (12 % busy) but the bytes a CU moves through its vector-memory path per output tile -- A tile + W tile in, C tile
// out, ~130 KB per 128 x 128 tile at ~50 GB/s per CU -- with a pipeline fill (operand latency) and a drain (stores) per
// tile that only co-resident workgroups overlap.  This kernel removes what it can of both:
//   * A-stationary row sweep: a workgroup owns ~M/512 rows (two workgroups per CU, every CU the same share: no tail
//     round), keeps their A block resident in LDS and sweeps ALL N / 128 column tiles over it; only W streams
//     (L2-resident, 48 KB per tile instead of 98 KB).
//   * one continuous global_load_lds ring over the W stages of all tiles: the next tile's first two stages are in flight
//     under the current tile's epilogue, and its stores are in flight under the next tile's first stages (counted vmcnt:
//     the waits behind a tile boundary leave the stores outstanding; the stages they guard were already retired by the
//     epilogue's own load wait).
// Waves: 4, side by side along N (32 columns each), every wave covers all 7 row groups of the block: accumulators 7 x 2
// tiles, A fragments from the resident block, W fragments from the ring.  Epilogue = gemm_common.h (bias, ReLU,
// dropout, gate), register-only.
#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int SW_THREADS = 256, SW_BN = 128, SW_MT = 7, SW_ROWS = 16 * SW_MT, SW_NS = 4, SW_DIST = SW_NS - 1, SW_BK = 32;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int sswz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}: gemm_nt.hip

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

// Wait until at most n entries of this wave's vector-memory queue are outstanding; n is wave-uniform and one of the sums
// {0,1,2} x PS + {0, MT stores} + {0, MT gate loads} the sweep produces (anything else drains the queue: always safe).
// (The builtin, not inline asm: the compiler's own waitcnt pass reads it and adds nothing of its own.)
__device__ __forceinline__ void wait_vm(int n) {
#define SW_W(k) case k: __builtin_amdgcn_s_waitcnt(vmcnt_imm(k)); break;
  switch (n) {
    SW_W(2) SW_W(4) SW_W(6) SW_W(7) SW_W(9) SW_W(11) SW_W(13) SW_W(14) SW_W(16) SW_W(18) SW_W(20)
    default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
  }
#undef SW_W
}

template <int EPI>
__global__ __launch_bounds__(SW_THREADS, 2) void gemm_sweep_kernel(const GemmParams p, int nwg) {
  constexpr int MT = SW_MT, NT = 2;
  constexpr int WSTAGE = SW_BN * SW_BK * 2;            // 8 KiB: one W stage [128 n-rows][64 B]
  constexpr int ASLICE = SW_ROWS * SW_BK * 2;          // 7 KiB: one k-slice of the A block [112 rows][64 B]
  constexpr int PS = WSTAGE / 1024 / 4;                // DMA pieces per wave per W stage (2)
  constexpr int GL = (EPI & EPI_GATE) ? MT : 0;        // gate loads per wave per tile
  static_assert(SW_DIST == 3 && PS == 2, "wait_vm's table");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.K / SW_BK;                          // 4..6
  unsigned char* ring = smem + nk * ASLICE;
  float* bias_s = reinterpret_cast<float*>(ring + SW_NS * WSTAGE);

  // this workgroup's rows: an even share of M (98 or 99 rows at cfg B; the host guarantees 97..112, so each of the 7 row
  // groups holds a valid row and a wave issues exactly MT stores per tile), not aligned to anything
  const int r0 = (int)((long)blockIdx.x * p.M / nwg), r1 = (int)((long)(blockIdx.x + 1) * p.M / nwg);
  GemmParams q = p;
  q.M = r1;                                            // the epilogue's row bound

#ifdef IQ_GEMM_STAMPS
  unsigned long long tw = 0, tc = 0, te = 0, t_start, t_a, t_x, t_y;
#define SW_NOW(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
  SW_NOW(t_start);
#else
#define SW_NOW(v) do {} while (0)
#endif
  // ---- oldest entries of the queue: the dropout step, this thread's 4 bias values (-> LDS: the tails then need no
  //      vector-memory load at all) ------------------------------------------------------------------------------------
  EpiRegs<MT, NT, EPI> R;
  R.rng = p.drop_on ? rng_resolve(p.rng) : p.rng;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + min(tid * 4, p.N - 4));

  const int prow = lane >> 2, pch = lane & 3;
  // ---- A block -> LDS once: nk slices x 7 pieces of 16 rows (rows past the share are clamped: never stored) ---------
  for (int pc = wave; pc < nk * MT; pc += 4) {
    const int ks = pc / MT, rg = pc - ks * MT;
    const int row = rg * 16 + prow;
    const int gm = min(r0 + row, r1 - 1);
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.A + (long)gm * p.lda + ks * SW_BK + (pch ^ sswz64(row)) * 8),
                                     (lds_void_t*)(smem + ks * ASLICE + rg * 1024), 16, 0, 0);
  }
  // ---- W stream: stage s = tile (s / nk), k-slice (s % nk) ------------------------------------------------------------
  // (N may end in a half tile -- 576 = 4.5 x 128: its missing W rows are clamped; the waves that own them skip the tail)
  const int ntile = (p.N + SW_BN - 1) / SW_BN, nstage = ntile * nk;
  // Sweep order rotated per workgroup, so that the workgroups, which run in step, do not all ask for the same W bytes.
  const int rot = blockIdx.x % ntile;
  auto tile_of = [&](int tq) { const int x = tq + rot; return x >= ntile ? x - ntile : x; };
  auto issue = [&](int s) {
    const int tq = s / nk, ks = s - tq * nk;
    const int t = tile_of(tq);
    unsigned char* st = ring + (s % SW_NS) * WSTAGE;
#pragma unroll
    for (int i = 0; i < PS; ++i) {
      const int row = (wave * PS + i) * 16 + prow;
      const int gn = min(t * SW_BN + row, p.N - 1);
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.B + (long)gn * p.ldb + ks * SW_BK + (pch ^ sswz64(row)) * 8),
                                       (lds_void_t*)(st + (wave * PS + i) * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < SW_DIST; ++i) issue(i);          // (nstage >= 2 * 4)

  // The A block and the bias are the only LDS this workgroup shares: once they have landed there is no further barrier.
  // A wave DMAs exactly the W rows it multiplies (pieces 2w, 2w+1 of a stage = its 32 columns), so ring slots, counted
  // waits and slot reuse are wave-private and the four waves drift apart: one wave's tail runs under the others' MFMAs.
  if (tid * 4 < p.N) *reinterpret_cast<f32x4*>(bias_s + tid * 4) = bias4;
  wait_vm(SW_DIST * PS);
  __builtin_amdgcn_s_barrier();
#ifdef IQ_GEMM_STAMPS
  SW_NOW(t_a);
#endif
  f32x4 acc[MT][NT];
  const int ch = lane >> 4, g = lane >> 4, c16 = lane & 15;
  const int lcol = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;   // this lane's 8 columns inside the wave's 32 (gemm_common.h)
  int s = 0, prev_st = 0;
  for (int t = 0; t < ntile; ++t) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col0 = tile_of(t) * SW_BN + wave * 32;
    const bool cols_ok = col0 < p.N;                     // wave-uniform: N % 64 == 0
    const int cur_gl = cols_ok ? GL : 0;
    for (int ks = 0; ks < nk; ++ks) {
      // Stage s landed?  Entries of this wave's queue younger than its DMAs: the y stages issued after it; for the first
      // DIST stages behind a tile boundary the previous tile's stores (which must not be waited for); for the last DIST
      // stages of a tile its gate loads.  All three are exact counts.
      SW_NOW(t_x);
      {
        int n = min(SW_DIST - 1, nstage - 1 - s) * PS;
        if (t > 0 && ks < SW_DIST) n += prev_st;
        if (GL && ks > nk - 1 - SW_DIST) n += cur_gl;
        wait_vm(n);
      }
      SW_NOW(t_y);
#ifdef IQ_GEMM_STAMPS
      tw += t_y - t_x;
#endif
      asm volatile("" ::: "memory");
      if (s + SW_DIST < nstage) issue(s + SW_DIST);
      if (GL && ks == nk - 1 - SW_DIST && cols_ok) {     // the tail's gate rows: a full ring of stages ahead of their use
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const long gm = min(r0 + i * 16 + c16, r1 - 1);
          R.gt[(EPI & EPI_GATE) ? i : 0][0] = *reinterpret_cast<const bf16x8*>(p.gate + gm * p.ldg + col0 + lcol);
        }
        asm volatile("" ::: "memory");
      }
      const bf16* As = reinterpret_cast<const bf16*>(smem + ks * ASLICE);
      const bf16* Bs = reinterpret_cast<const bf16*>(ring + (s % SW_NS) * WSTAGE);
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * SW_BK + (ch ^ sswz64(row)) * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wave * 32 + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * SW_BK + (ch ^ sswz64(row)) * 8);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: gemm_common.h
      ++s;
#ifdef IQ_GEMM_STAMPS
      asm volatile("s_nop 0" :: "v"(acc[MT - 1][NT - 1]));     // (the last MFMA has retired)
      SW_NOW(t_x);
      tc += t_x - t_y;
#endif
    }
    // tail of tile t: bias from LDS, gate rows waited for with the next tile's stages left in flight; the stores stay in
    // flight under the next tile's first stages
    SW_NOW(t_x);
    prev_st = 0;
    if (cols_ok) {
      R.bias_lo[0] = *reinterpret_cast<const f32x4*>(bias_s + col0 + lcol);
      R.bias_hi[0] = *reinterpret_cast<const f32x4*>(bias_s + col0 + lcol + 4);
      if (GL) wait_vm(min(SW_DIST, nstage - s) * PS);
      epi_finish<MT, NT, EPI>(q, acc, R, r0, col0, lane);
      prev_st = MT;
    }
#ifdef IQ_GEMM_STAMPS
    SW_NOW(t_y);
    te += t_y - t_x;
#endif
  }
#ifdef IQ_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SW_NOW(t_x);
  if (lane == 0 && p.stamps) {
    unsigned long long* o = p.stamps + ((long)blockIdx.x * 4 + wave) * 6;
    o[0] = t_start; o[1] = t_a; o[2] = tw; o[3] = tc; o[4] = te; o[5] = t_x;
  }
#endif
}

}  // namespace

// Shapes this kernel is launched for (iq_gemm_bf16_nt asks; everything else goes to its tiled kernels): wide N in
// 64-column units, a contraction short enough for the A block to stay resident beside a second workgroup (K <= 192), and
// a row count whose even 512-way split fits the 112-row block and fills most of it.
static int sweep_workgroups(int M, int N, int K) {
  if (N % 64 != 0 || N < 256 || N > 1024 || K % SW_BK != 0 || K < 128 || K > 192) return 0;
  int nwg = 512;
  while ((M + nwg - 1) / nwg > SW_ROWS) nwg += 512;                 // larger M: whole further rounds of the same shape
  return M / nwg >= SW_ROWS - 15 ? nwg : 0;                         // every one of the 7 row groups must hold a valid row
}

extern "C" int iq_gemm_sweep_supported(int M, int N, int K) { return sweep_workgroups(M, N, K) > 0 ? 1 : 0; }

// Called by iq_gemm_bf16_nt with its resolved parameters (bias non-null).  Returns false when the shape / epilogue is not
// this kernel's; true after a launch.
bool gemm_sweep_try(const GemmParams& p, int epi_mode, hipStream_t st) {
  if (epi_mode != 0 && epi_mode != EPI_GATE) return false;
  const int nwg = sweep_workgroups(p.M, p.N, p.K);
  if (!nwg || (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C | (uintptr_t)p.bias) % 16)) return false;
  if (epi_mode == EPI_GATE && ((uintptr_t)p.gate % 16)) return false;
  const size_t lds = (size_t)(p.K / SW_BK) * (SW_ROWS * SW_BK * 2) + (size_t)SW_NS * (SW_BN * SW_BK * 2) + (size_t)p.N * 4;   // K = 192, N = 768: 42 + 32 + 3 KiB
  if (epi_mode == EPI_GATE) {
    auto k = gemm_sweep_kernel<EPI_GATE>;
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)attr;
    k<<<nwg, SW_THREADS, lds, st>>>(p, nwg);
  } else {
    auto k = gemm_sweep_kernel<0>;
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)attr;
    k<<<nwg, SW_THREADS, lds, st>>>(p, nwg);
  }
  return true;
}
