// Shared by the NT GEMM kernels: launch parameters and the register-only fused epilogue.
#pragma once
#include "common.h"

struct GemmParams {
#ifdef IQ_GEMM_STAMPS
  unsigned long long* stamps;   // diagnostic build: [grid][6] s_memtime values
#endif
  const bf16* A; const bf16* B; bf16* C;
  int lda, ldb, ldc, M, N, K;
  const float* bias; int relu;
  const float* pe; int tok, seq, cls_off;
  int drop_on; IqRng rng; uint32_t thresh; float dscale;
  const bf16* gate; int ldg; float gate_scale;
  const bf16* residual; int ldr;
  int tiles_m, tiles_n;
  int stagger;     // start-phase spread of co-resident workgroups, in units of ~512 cycles (0 = off)
};

// Epilogue shared by both kernels -- registers only, no LDS round trip.
// The MFMAs are issued with the WEIGHT fragment as the A operand, so an accumulator tile is C^T:
// lane (g = lane>>4, c = lane&15) holds C[row = 16i + c][cols 16j + 4g .. 4g+3].  One
// v_permlane16_swap per register between the two tiles of a column pair (2jp, 2jp+1) leaves every lane
// with EIGHT consecutive columns of one row (even g: tile 2jp, cols 4g..4g+7; odd g: tile 2jp+1, cols
// 4(g-1)..4(g-1)+7): one 16 B bf16 store per lane, a wave instruction covers 16 rows x 64 B, and the
// 8-column group is exactly the Philox dropout group, so the whole elementwise tail (bias, ReLU, PE,
// dropout, gate, residual) runs lane-locally on fp32 before the single rounding to bf16.
// (The first version staged fp32 strips through LDS: 45-55 % of a workgroup's life was this epilogue.)
// EPI bit 0: residual, bit 1: gate, bit 2: positional table (+ row remap).  Compile-time so that each
// variant is straight-line: every global LOAD of the tail (bias, residual, gate) is issued up front and
// retired by ONE counted wait, and the store loop then contains no vmcnt wait at all.  (CDNA counts
// loads and stores in one in-order vmcnt: a load waited for after a store also waits for that store, and
// the runtime-flag version of this loop serialised its 8 stores on the full write latency -- 8.2k of the
// 18k cycles a workgroup lived.)
constexpr int EPI_RES = 1, EPI_GATE = 2, EPI_PE = 4;

// row0 / col0: global row / column of the wave's first accumulator tile; the wave owns MT x NT tiles of 16x16.
// The tail is split in two (gemm_epilogue = epi_load + vmcnt(0) + epi_finish) so that a kernel may place the loads
// elsewhere.  Measured: issuing them BEFORE the K loop of the tiled kernel (so that the epilogue never waits) is slower
// (gemm_nt 3.32 vs 3.14 ms/step; gate variant 52 vs 45 us): vmcnt is in-order, so the first operand stage then waits
// behind 16-32 KB of residual / gate per workgroup.
template <int MT, int NT, int EPI>
struct EpiRegs {
  static constexpr int NP = NT / 2;
  f32x4 bias_lo[NP], bias_hi[NP];
  bf16x8 res[(EPI & EPI_RES) ? MT : 1][NP], gt[(EPI & EPI_GATE) ? MT : 1][NP];
  IqRng rng;      // resolved here: rng_resolve may LOAD the device-resident step, which must sit before the one wait
};

template <int MT, int NT, int EPI>
__device__ __forceinline__ void epi_load(const GemmParams& p, EpiRegs<MT, NT, EPI>& R, int row0, int col0, int lane) {
  static_assert(NT % 2 == 0, "column tiles are consumed in pairs");
  constexpr int NP = NT / 2;
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  R.rng = p.drop_on ? rng_resolve(p.rng) : p.rng;
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    const int col = col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g);
    R.bias_lo[jp] = f32x4{0.f, 0.f, 0.f, 0.f};
    R.bias_hi[jp] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && col < p.N) {
      R.bias_lo[jp] = *reinterpret_cast<const f32x4*>(p.bias + col);
      R.bias_hi[jp] = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = row0 + i * 16 + c16;
      const bool ok = gm < p.M && col < p.N;
      if (EPI & EPI_RES) {
        long orow = gm;
        if (EPI & EPI_PE) { const int f = gm / p.tok; orow = (long)f * p.seq + (gm - f * p.tok) + p.cls_off; }
        R.res[i][jp] = bf16x8{};
        if (ok) R.res[i][jp] = *reinterpret_cast<const bf16x8*>(p.residual + orow * p.ldr + col);
      }
      if (EPI & EPI_GATE) {
        R.gt[i][jp] = bf16x8{};
        if (ok) R.gt[i][jp] = *reinterpret_cast<const bf16x8*>(p.gate + (long)gm * p.ldg + col);
      }
    }
  }
}

// The tail's loads with NO predicate: rows / columns outside the problem are clamped to valid addresses (their values are
// never stored) and `p.bias` must be non-null (the host passes a zero vector when the caller gave none), so the number
// of vector-memory instructions is the compile-time constant EPI_EARLY_LOADS.  A kernel can then issue them behind its
// last operand stage and keep waiting for that stage with a COUNTED vmcnt: they fly under the last MFMAs instead of
// after them (tools/check at build time: scripts/dbg/check_epi_counts.py counts them in the ISA).  R.rng is not touched.
template <int MT, int NT, int EPI>
constexpr int epi_early_loads() {
  return 2 * (NT / 2) + MT * (NT / 2) * (((EPI & EPI_RES) ? 1 : 0) + ((EPI & EPI_GATE) ? 1 : 0));
}
template <int MT, int NT, int EPI>
__device__ __forceinline__ void epi_load_early(const GemmParams& p, EpiRegs<MT, NT, EPI>& R, int row0, int col0, int lane) {
  static_assert(!((EPI & EPI_RES) && (EPI & EPI_PE)), "the row-remapped residual is not used by any caller");
  constexpr int NP = NT / 2;
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    int col = col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g);
    col = col < p.N ? col : 0;
    R.bias_lo[jp] = *reinterpret_cast<const f32x4*>(p.bias + col);
    R.bias_hi[jp] = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const long gm = min(row0 + i * 16 + c16, p.M - 1);
      if (EPI & EPI_RES) R.res[i][jp] = *reinterpret_cast<const bf16x8*>(p.residual + gm * p.ldr + col);
      if (EPI & EPI_GATE) R.gt[i][jp] = *reinterpret_cast<const bf16x8*>(p.gate + gm * p.ldg + col);
    }
  }
}

// LDSOUT (chained GEMMs, gemm_chain.hip): the rounded bf16 tile is ALSO left in LDS as the A operand of the next
// product: [rows][256 B] image at `hs`, 16 B chunk c of row r at chunk c ^ (r & 15); (lrow0, lcol0) = the wave tile's
// origin inside that image.
template <int MT, int NT, int EPI, bool LDSOUT = false>
__device__ __forceinline__ void epi_finish(const GemmParams& p, f32x4 (&acc)[MT][NT], const EpiRegs<MT, NT, EPI>& R, int row0,
                                           int col0, int lane, unsigned char* hs = nullptr, int lrow0 = 0, int lcol0 = 0) {
#ifdef IQ_EPI_SKIP   // ablation build: timing only
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(acc[i][j]));
  return;
#endif
  constexpr int NP = NT / 2;       // column-tile pairs
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const IqRng rng = R.rng;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = row0 + i * 16 + c16;
    long orow = gm;
    int prow = 0;
    if (EPI & EPI_PE) {
      const int f = gm / p.tok, tk = gm - f * p.tok;
      prow = tk + p.cls_off;
      orow = (long)f * p.seq + prow;
    }
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      float w[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (copy the vector elements to scalars first: bit-casting an ext-vector element lvalue made clang
        //  read element 0 for every r)
        const float va = acc[i][2 * jp][r], vb = acc[i][2 * jp + 1][r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        w[r] = __uint_as_float(sw[0]) + R.bias_lo[jp][r];
        w[4 + r] = __uint_as_float(sw[1]) + R.bias_hi[jp][r];
      }
      const int col = col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g);
      if (gm < p.M && col < p.N) {
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = fmaxf(w[e], 0.f);
        }
        if (EPI & EPI_PE) {
          const f32x4 pa = *reinterpret_cast<const f32x4*>(p.pe + (long)prow * p.N + col);
          const f32x4 pb = *reinterpret_cast<const f32x4*>(p.pe + (long)prow * p.N + col + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[e] += pa[e]; w[4 + e] += pb[e]; }
        }
        if (p.drop_on) {
          const uint32_t keep = dropout_keep8(rng, (uint64_t)(orow * p.N + col) >> 3, p.thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale : 0.f;
        }
        if (EPI & EPI_GATE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = ((float)R.gt[i][jp][e] > 0.f) ? w[e] * p.gate_scale : 0.f;
        }
        if (EPI & EPI_RES) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] += (float)R.res[i][jp][e];
        }
        const bf16x8 packed = pack8(w);
        if (LDSOUT) {
          const int lr = lrow0 + i * 16 + c16, lc = (lcol0 + (col - col0)) >> 3;
          *reinterpret_cast<bf16x8*>(hs + lr * 256 + ((lc ^ (lr & 15)) << 4)) = packed;
        }
#ifdef IQ_EPI_NO_STORE
        if (p.ldc < 0)
#endif
        *reinterpret_cast<bf16x8*>(p.C + orow * p.ldc + col) = packed;
      }
    }
  }
}

template <int MT, int NT, int EPI, bool LDSOUT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MT][NT], int row0, int col0, int lane,
                                              unsigned char* hs = nullptr, int lrow0 = 0, int lcol0 = 0) {
  EpiRegs<MT, NT, EPI> R;
  epi_load<MT, NT, EPI>(p, R, row0, col0, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): every load above has landed; none below (PE variant excepted)
  epi_finish<MT, NT, EPI, LDSOUT>(p, acc, R, row0, col0, lane, hs, lrow0, lcol0);
}

// gemm_big.hip: 256 x 256 tiles, persistent, for the MFMA-bound shapes (N % 256 == 0, K >= 256, enough tiles to fill the
// chip twice).  Launches and returns true when the shape and epilogue are its own.
bool gemm_big_try(const GemmParams& p, int epi_mode, hipStream_t st);
