// The position-wise feed-forward sub-layer of one FRAME in one workgroup (gfx950):
//
//     H[S,F]  = dropout1(relu(X1[S,D] * W1[F,D]^T + b1))                      bf16, written once (backward reads it)
//     Z[S,D]  = dropout2(H * W2[D,F]^T + b2) + X1                              bf16 (kept for backward)
//     X[S,D]  = gamma * (Z - mean) * rstd + beta ; mean, rstd fp32 [S]         (norm2, eps inside the square root)
//
// = PositionwiseFeedForward.forward (V/models/layers/position_wise_feed_forward.py:12-17: Linear, ReLU, Dropout, Linear)
// followed by `x = norm2(dropout2(ffn(x)) + x)` of EncoderLayer.forward (V/models/blocks/encoder_layer.py:30-33).
//
// Run as two launches (FFN1 GEMM, then FFN2 GEMM + LayerNorm) the hidden activation H -- 4x the width of every other
// activation -- is written to HBM and read back at once (cfg B: 77.5 MB each way per layer), and each launch pays its own
// pipeline fill and drain.  Here a workgroup owns the rows of one frame (cfg B: 197 rows = 13 row groups of 16, one frame
// per CU), keeps X1 in LDS for the whole kernel, and walks the hidden dimension in chunks of 128 columns:
//     phase 1   H_c = relu(X1 W1_c^T + b1_c), dropout1  -> global memory (for backward) AND an LDS image
//     phase 2   acc2 += H_c W2_c^T                        (fp32 accumulators: rows of the wave x 48 / 32 columns)
// so H crosses HBM once, written.  Weight fragments go from L2 straight to registers (every CU streams the same 0.6 MB per
// layer; no ring, no DMA, no swizzled staging): the only LDS traffic is the activation side of the MFMAs.
// Waves: 8 = 4 column groups x 2 row halves.  Phase 1: 32 hidden columns (2 tiles) x 7 row groups per wave; phase 2:
// D/4 output columns (3 | 2 tiles) x 7 row groups.  Same MFMA orientation and register-only tail as gemm_common.h:
// weight fragment as the A operand, so lane (g, c) holds C[row 16 i + c][cols 16 j + 4 g .. + 3], and one
// v_permlane16_swap per register gives a lane 8 consecutive columns = one Philox dropout group = one 16-byte store.
// K order of both products is the plain ascending one: H and Z are bit-identical to iq_gemm_bf16_nt + iq_gemm_bf16_ln.
// Tail: Z (rounded to bf16) replaces X1 in its LDS image in place, then LayerNorm runs row-wise on the image (8 lanes per
// row, ln_fwd_kernel's two-pass arithmetic) and writes Z, X and the statistics with coalesced 16-byte stores.
#include <stdlib.h>

#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int FC_THREADS = 256, FC_CHUNK = 128, FC_RGW = 7, FC_MAXROWS = 2 * FC_RGW * 16;   // 224 rows: 14 row groups in two halves

struct FfnChainParams {
  const bf16* X1; const bf16* W1; const bf16* W2;      // [M,D], [F,D], [D,F]
  const float* b1; const float* b2; const float* gamma; const float* beta;
  bf16* H; bf16* Z; bf16* X; float* mean; float* rstd;  // [M,F], [M,D], [M,D], [M], [M]
  int S, F;                                             // rows per frame (= per workgroup), hidden width
  float eps;
  int drop1_on, drop2_on; IqRng rng1, rng2; uint32_t thresh1, thresh2; float dscale1, dscale2;
};

// 16-byte chunk `ch` of row `row` in an image with CPR chunks per row: rows of 256 B (CPR 16) XOR the chunk with row & 15,
// rows of 384 B (CPR 24; 24 = 8 mod 16, so the row's parity already moves the slot by 8) XOR its low 3 bits with (row >> 1) & 7:
// the 16 rows x {k-chunk g, g+1} a ds_read_b128 lane group touches land on 16 distinct 16-byte slots of the 256-byte bank row.
template <int CPR> __device__ __forceinline__ int fc_swz(int row, int ch) {
  return CPR == 16 ? (ch ^ (row & 15)) : ((ch & ~7) | ((ch & 7) ^ ((row >> 1) & 7)));
}

template <int D>
__global__ __launch_bounds__(FC_THREADS, 2) void ffn_chain_fwd_kernel(const FfnChainParams p) {
  constexpr int XCPR = D / 8, HCPR = FC_CHUNK / 8;      // chunks per row: X1 image 24 | 16, H chunk image 16
  constexpr int CT2 = D / 64;                           // output column tiles per wave in phase 2: 3 | 2
  constexpr int KS1 = D / 32, KS2 = FC_CHUNK / 32;      // k-steps: 6 | 4, 4
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cp = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = column group
  const int rh = blockIdx.x & 1;                              // workgroup = one row half of a frame
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const int S = p.S, F = p.F;
  const int nrg = (S + 15) >> 4;                        // row groups of the frame (<= 14)
  const int cnt0 = (nrg + 1) >> 1;                      // row half 0 takes the larger share
  const int rg0 = rh == 0 ? 0 : cnt0;
  const int cnt = rh == 0 ? cnt0 : nrg - cnt0;          // this wave's row groups (<= 7), wave-uniform
  const long row_base = (long)(blockIdx.x >> 1) * S;    // first global row of the frame
  const int lrow0 = rg0 * 16;                           // first frame row of this half; image rows are relative to it
  const int hrows = cnt * 16;                           // image rows of this half
  unsigned char* XI = smem;                             // [cnt0 * 16][D * 2 B], swizzled
  unsigned char* HC = smem + (size_t)cnt0 * 16 * D * 2; // [cnt0 * 16][256 B], swizzled
  if (cnt == 0) return;                                 // (a one-group frame has no second half; uniform)

  // ---- stage X1 (contiguous S x D block) into its image; rows >= S are zero ------------------------------------------------
  {
    const bf16* src = p.X1 + (row_base + lrow0) * D;
    const int total = hrows * XCPR, live = (min(S, lrow0 + hrows) - lrow0) * XCPR;
    for (int base = 0; base < total; base += 4 * FC_THREADS) {
      bf16x8 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const bf16x8*>(src + (long)min(base + j * FC_THREADS + tid, live - 1) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int id = base + j * FC_THREADS + tid;
        if (id < total) {
          const int r = id / XCPR, c = id - r * XCPR;
          *reinterpret_cast<bf16x8*>(XI + r * (D * 2) + fc_swz<XCPR>(r, c) * 16) = id < live ? v[j] : bf16x8{};
        }
      }
    }
  }
  const IqRng rng1 = p.drop1_on ? rng_resolve(p.rng1) : p.rng1;
  const IqRng rng2 = p.drop2_on ? rng_resolve(p.rng2) : p.rng2;
  __syncthreads();

  f32x4 acc2[FC_RGW][CT2];
#pragma unroll
  for (int i = 0; i < FC_RGW; ++i)
#pragma unroll
    for (int j = 0; j < CT2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = F / FC_CHUNK;
  // this lane's weight-fragment rows: phase 1 hidden unit (chunk base + 32 cp + 16 ct + c16), phase 2 output column (D/4 cp + 16 ct + c16)
  const bf16* w1_lane = p.W1 + (long)(32 * cp + c16) * D + 8 * g;
  const bf16* w2_lane = p.W2 + (long)((D / 4) * cp + c16) * F + 8 * g;
  // Weight fragments are requested PD k-steps ahead of their MFMAs (program order pinned by the fences below): holding a whole
  // chunk's worth (48 + 48 registers beside 140 accumulators) spilled.
  constexpr int PD = 2, PD2 = 1;   // W1 two k-steps ahead, W2 one (two of each spills 5 registers at D = 192)
  bf16x8 w1f[2][KS1];
  auto load_w1 = [&](int f0, int ks) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) w1f[ct][ks] = *reinterpret_cast<const bf16x8*>(w1_lane + (long)(f0 + 16 * ct) * D + 32 * ks);
  };
#pragma unroll
  for (int ks = 0; ks < PD; ++ks) load_w1(0, ks);

  for (int c = 0; c < nchunk; ++c) {
    const int f0 = c * FC_CHUNK;
    // ---- phase 1: H_c = relu(X1 W1_c^T + b1_c), dropout1 ---------------------------------------------------------------------
    f32x4 acc1[FC_RGW][2];
#pragma unroll
    for (int i = 0; i < FC_RGW; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      if (ks + PD < KS1) load_w1(f0, ks + PD);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < FC_RGW; ++i) {
        if (i < cnt) {
          const int row = i * 16 + c16;
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(XI + row * (D * 2) + fc_swz<XCPR>(row, 4 * ks + g) * 16);
          acc1[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0][ks], af, acc1[i][0], 0, 0, 0);
          acc1[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1][ks], af, acc1[i][1], 0, 0, 0);
        }
      }
    }
    // the first W2 fragments of the chunk travel under the epilogue below
    bf16x8 w2f[CT2][KS2];
    auto load_w2 = [&](int ks) {
#pragma unroll
      for (int ct = 0; ct < CT2; ++ct) w2f[ct][ks] = *reinterpret_cast<const bf16x8*>(w2_lane + (long)(16 * ct) * F + f0 + 32 * ks);
    };
#pragma unroll
    for (int ks = 0; ks < PD2; ++ks) load_w2(ks);
    asm volatile("" ::: "memory");
    {
      const int col = f0 + 32 * cp + (odd ? 16 + 4 * (g - 1) : 4 * g);       // first of this lane's 8 hidden columns
      const f32x4 b_lo = *reinterpret_cast<const f32x4*>(p.b1 + col), b_hi = *reinterpret_cast<const f32x4*>(p.b1 + col + 4);
      const int hch = (col - f0) >> 3;
#pragma unroll
      for (int i = 0; i < FC_RGW; ++i) {
        if (i < cnt) {
          const int row = i * 16 + c16;
          float w[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float va = acc1[i][0][r], vb = acc1[i][1][r];
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            w[r] = fmaxf(__uint_as_float(sw[0]) + b_lo[r], 0.f);
            w[4 + r] = fmaxf(__uint_as_float(sw[1]) + b_hi[r], 0.f);
          }
          const long grow = row_base + lrow0 + row;
          if (p.drop1_on) {
            const uint32_t keep = dropout_keep8(rng1, (uint64_t)(grow * F + col) >> 3, p.thresh1);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale1 : 0.f;
          }
          const bf16x8 hb = pack8(w);
          *reinterpret_cast<bf16x8*>(HC + row * 256 + fc_swz<HCPR>(row, hch) * 16) = hb;
          if (lrow0 + row < S) *reinterpret_cast<bf16x8*>(p.H + grow * F + col) = hb;
        }
      }
    }
    __syncthreads();                                    // H_c complete in LDS
    // ---- phase 2: acc2 += H_c W2_c^T; the next chunk's first W1 fragments are requested behind its last W2 ones -------------
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      if (ks + PD2 < KS2) load_w2(ks + PD2);
      if (ks + PD >= KS2 && c + 1 < nchunk) load_w1(f0 + FC_CHUNK, ks + PD - KS2);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < FC_RGW; ++i) {
        if (i < cnt) {
          const int row = i * 16 + c16;
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(HC + row * 256 + fc_swz<HCPR>(row, 4 * ks + g) * 16);
#pragma unroll
          for (int ct = 0; ct < CT2; ++ct) acc2[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ct][ks], af, acc2[i][ct], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                    // every wave has read H_c: the image may be overwritten
  }

  // ---- tail 1: z = dropout2(acc2 + b2) + x1, rounded to bf16, replaces x1 in its image ------------------------------------------
  // column pairs through one permlane swap (8 columns per lane); D = 192 leaves a third tile per wave: 4 columns per lane,
  // its Philox group shared with the lane pair g ^ 1 (both draw the same 8 flags and use their half)
  {
    const int n0 = (D / 4) * cp;
#pragma unroll
    for (int i = 0; i < FC_RGW; ++i) {
      if (i < cnt) {
        const int row = i * 16 + c16;
        const long grow = row_base + lrow0 + row;
        {
          const int col = n0 + (odd ? 16 + 4 * (g - 1) : 4 * g);
          const f32x4 b_lo = *reinterpret_cast<const f32x4*>(p.b2 + col), b_hi = *reinterpret_cast<const f32x4*>(p.b2 + col + 4);
          float w[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float va = acc2[i][0][r], vb = acc2[i][1][r];
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            w[r] = __uint_as_float(sw[0]) + b_lo[r];
            w[4 + r] = __uint_as_float(sw[1]) + b_hi[r];
          }
          if (p.drop2_on) {
            const uint32_t keep = dropout_keep8(rng2, (uint64_t)(grow * D + col) >> 3, p.thresh2);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale2 : 0.f;
          }
          bf16x8* slot = reinterpret_cast<bf16x8*>(XI + row * (D * 2) + fc_swz<XCPR>(row, col >> 3) * 16);
          const bf16x8 res = *slot;
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] += (float)res[e];
          *slot = pack8(w);
        }
        if constexpr (CT2 == 3) {
          const int col = n0 + 32 + 4 * g;                 // this lane's 4 columns of the third tile
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b2 + col);
          float w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = acc2[i][2][r] + b4[r];
          if (p.drop2_on) {
            const uint32_t keep = dropout_keep8(rng2, (uint64_t)(grow * D + col) >> 3, p.thresh2) >> (4 * (g & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale2 : 0.f;
          }
          bf16x4* slot = reinterpret_cast<bf16x4*>(XI + row * (D * 2) + fc_swz<XCPR>(row, col >> 3) * 16 + (col & 7) * 2);
          const bf16x4 res = *slot;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(w[e] + (float)res[e]);
          *slot = o;
        }
      }
    }
  }
  __syncthreads();
  // ---- tail 2: LayerNorm over the image rows: 8 lanes per row, NV 16-byte vectors per lane (ln_fwd_kernel's arithmetic) ---------
  {
    constexpr int LPR = 8, NV = XCPR / LPR;                // 3 | 2
    const int lj = tid & 7, rsub = tid >> 3;               // 32 rows per pass
    const float invD = 1.0f / (float)D;
    const int nrows = min(S, lrow0 + hrows) - lrow0;       // real rows of this half
    for (int r0 = 0; r0 < nrows; r0 += FC_THREADS / LPR) {
      const int row = r0 + rsub;
      if (row < nrows) {                                    // (a row's 8 lanes are all in or all out)
        float z[NV][8];
        float s1 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const bf16x8 zb = *reinterpret_cast<const bf16x8*>(XI + row * (D * 2) + fc_swz<XCPR>(row, v * LPR + lj) * 16);
          unpack8(zb, z[v]);
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += z[v][e];
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 * invD;
        float s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = z[v][e] - mean; s2 += d * d; }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = 1.0f / sqrtf(s2 * invD + p.eps);
        const long grow = row_base + lrow0 + row;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int col = (v * LPR + lj) * 8;
          const f32x4 g_lo = *reinterpret_cast<const f32x4*>(p.gamma + col), g_hi = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
          const f32x4 b_lo = *reinterpret_cast<const f32x4*>(p.beta + col), b_hi = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
          float y[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = g_lo[e] * ((z[v][e] - mean) * rstd) + b_lo[e];
            y[4 + e] = g_hi[e] * ((z[v][4 + e] - mean) * rstd) + b_hi[e];
          }
          *reinterpret_cast<bf16x8*>(p.Z + grow * D + col) = pack8(z[v]);
          *reinterpret_cast<bf16x8*>(p.X + grow * D + col) = pack8(y);
        }
        if (lj == 0) { p.mean[grow] = mean; p.rstd[grow] = rstd; }
      }
    }
  }
}

template <int D>
int launch_chain(const FfnChainParams& p, int frames, hipStream_t st) {
  const int nrg = (p.S + 15) / 16;
  const size_t lds = (size_t)((nrg + 1) / 2) * 16 * (D * 2 + 256);
  auto k = ffn_chain_fwd_kernel<D>;
  static size_t attr_set = 0;
  if (lds > 48 * 1024 && lds > attr_set) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = lds;
  }
  k<<<2 * frames, FC_THREADS, lds, st>>>(p);
  return iq_launch_status();
}

}  // namespace

// D = 128 | 192, F a multiple of 128, at most 224 rows per frame (both images of a frame fit one CU's LDS)
extern "C" int iq_ffn_chain_supported(int S, int D, int F) {
  if (!(D == 128 || D == 192) || F < FC_CHUNK || (F % FC_CHUNK) || S <= 0 || S > FC_MAXROWS) return 0;
  return 1;
}

extern "C" int iq_ffn_chain_fwd(const void* X1, const void* W1, const float* b1, const iq_dropout_t* drop1, void* H,
                                const void* W2, const float* b2, const iq_dropout_t* drop2, const float* gamma,
                                const float* beta, float eps, void* Z, void* X, float* mean, float* rstd, int frames, int S,
                                int D, int F, iq_stream_t stream) {
  if (frames <= 0) return IQ_OK;
  if (!X1 || !W1 || !b1 || !H || !W2 || !b2 || !gamma || !beta || !Z || !X || !mean || !rstd) return IQ_ERR_ARG;
  if (!iq_ffn_chain_supported(S, D, F)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)X1 | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)H | (uintptr_t)Z | (uintptr_t)X | (uintptr_t)b1 | (uintptr_t)b2 |
       (uintptr_t)gamma | (uintptr_t)beta) % 16) return IQ_ERR_ARG;
  FfnChainParams p = {};
  p.X1 = (const bf16*)X1; p.W1 = (const bf16*)W1; p.W2 = (const bf16*)W2;
  p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta;
  p.H = (bf16*)H; p.Z = (bf16*)Z; p.X = (bf16*)X; p.mean = mean; p.rstd = rstd;
  p.S = S; p.F = F; p.eps = eps;
  auto fill = [](const iq_dropout_t* d, int* on, IqRng* r, uint32_t* th, float* sc) -> bool {
    *on = 0; *th = 0; *sc = 1.f; *r = IqRng{0, 0, 0, nullptr};
    if (d && d->p > 0.f) {
      if (d->p >= 1.f) return false;
      *on = 1;
      r->seed = d->seed; r->step = d->step; r->site = d->site; r->step_dev = d->step_dev;
      *th = dropout_thresh(d->p);
      *sc = dropout_scale(d->p);
    }
    return true;
  };
  if (!fill(drop1, &p.drop1_on, &p.rng1, &p.thresh1, &p.dscale1) || !fill(drop2, &p.drop2_on, &p.rng2, &p.thresh2, &p.dscale2))
    return IQ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  const double M = (double)frames * S;
  IQ_PROF_K(2.0 * (M * D * 3 + M * F + 2.0 * D * F) + 8.0 * M, 4.0 * M * D * F, "ffn_chain_fwd_kernel<%d>", D);
  return D == 192 ? launch_chain<192>(p, frames, st) : launch_chain<128>(p, frames, st);
}
