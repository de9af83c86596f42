// LayerNorm forward / backward for the post-norm encoder (gfx950).
//
// Reference: Transformer_Thesis/ViT/models/layers/layers_norm.py:11-19
//   mean, biased var over the last dim, (x-mean)/sqrt(var+eps), gamma*out+beta, eps=1e-12.
// Storage bf16, statistics and normalisation fp32 (eps=1e-12 is invisible in bf16).
//
// HBM-bound: fwd moves 2*M*D*2 B, bwd 3..4*M*D*2 B.  A row is owned by LPR lanes of a wave
// (LPR in {1..64}, power of two), each lane holding NV 16-byte vectors, so D = LPR*NV*8 and
// every global access is a 16 B/lane coalesced vector (guide G13).  No LDS in fwd.
#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int LN_THREADS = 256;

template <int LPR, int NV, bool MASK = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const bf16* __restrict__ Z, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ X,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int M, int D, float eps) {
  constexpr int RPW = 64 / LPR;                 // rows per wave
  constexpr int RPB = RPW * (LN_THREADS / 64);  // rows per block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % LPR, rsub = lane / LPR;
  float g[NV][8], b[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + j) * 8;
    const bool cv = !MASK || c < D;
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[v][e] = cv ? gamma[c + e] : 0.f; b[v][e] = cv ? beta[c + e] : 0.f; }
  }
  const float invD = 1.0f / (float)D;
  for (long row0 = (long)blockIdx.x * RPB; row0 < M; row0 += (long)gridDim.x * RPB) {
    const long row = row0 + wave * RPW + rsub;
    const bool ok = row < M;
    float x[NV][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      bf16x8 t = {};
      if (ok && (!MASK || (v * LPR + j) * 8 < D)) t = *reinterpret_cast<const bf16x8*>(Z + row * D + (v * LPR + j) * 8);
      unpack8(t, x[v]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[v][e];
    }
    const float mean = group_sum<LPR>(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = x[v][e] - mean; q += (!MASK || (v * LPR + j) * 8 < D) ? d * d : 0.f; }
    const float var = group_sum<LPR>(q) * invD;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (ok) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = g[v][e] * ((x[v][e] - mean) * rstd) + b[v][e];
        if (!MASK || (v * LPR + j) * 8 < D) *reinterpret_cast<bf16x8*>(X + row * D + (v * LPR + j) * 8) = pack8(y);
      }
      if (j == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
  }
}

// Backward.  dZ = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dX*gamma.
// Optionally also emits dY = dropout_mask(dZ)*scale, the gradient w.r.t. the GEMM output that
// was dropped out before the residual add (encoder_layer.py:24-25,32-33); the mask is
// regenerated from the Philox counter, never stored.
// dgamma/dbeta: per-lane register accumulation over the block's rows -> LDS -> one partial row
// per block; ln_bwd_reduce_kernel sums the partials (deterministic, no atomics).
template <int LPR, int NV, bool DROP, bool MASK = false>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const bf16* __restrict__ dX, const bf16* __restrict__ Z,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, bf16* __restrict__ dZ,
                                                            bf16* __restrict__ dY, IqRng rng, uint32_t thresh, float dscale,
                                                            float* __restrict__ partial, int M, int D) {
  constexpr int RPW = 64 / LPR;
  constexpr int RPB = RPW * (LN_THREADS / 64);
  extern __shared__ __attribute__((aligned(16))) float red[];  // [RPB][2*D] floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % LPR, rsub = lane / LPR;
  float g[NV][8], ag[NV][8], ab[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + j) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[v][e] = (!MASK || c < D) ? gamma[c + e] : 0.f; ag[v][e] = 0.f; ab[v][e] = 0.f; }
  }
  const float invD = 1.0f / (float)D;
  if (DROP) rng = rng_resolve(rng);
  for (long row0 = (long)blockIdx.x * RPB; row0 < M; row0 += (long)gridDim.x * RPB) {
    const long row = row0 + wave * RPW + rsub;
    const bool ok = row < M;
    float xh[NV][8], dy[NV][8];
    float mean = 0.f, rstd = 0.f;
    if (ok) { mean = mean_in[row]; rstd = rstd_in[row]; }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      bf16x8 tz = {}, td = {};
      const bool cv = !MASK || (v * LPR + j) * 8 < D;
      if (ok && cv) {
        tz = *reinterpret_cast<const bf16x8*>(Z + row * D + (v * LPR + j) * 8);
        td = *reinterpret_cast<const bf16x8*>(dX + row * D + (v * LPR + j) * 8);
      }
      unpack8(tz, xh[v]);
      unpack8(td, dy[v]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[v][e] = cv ? (xh[v][e] - mean) * rstd : 0.f;
        ag[v][e] += dy[v][e] * xh[v][e];
        ab[v][e] += dy[v][e];
        dy[v][e] *= g[v][e];
        s1 += dy[v][e];
        s2 += dy[v][e] * xh[v][e];
      }
    }
    const float c1 = group_sum<LPR>(s1) * invD;
    const float c2 = group_sum<LPR>(s2) * invD;
    if (ok) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (dy[v][e] - c1 - xh[v][e] * c2);
        const long off = row * D + (v * LPR + j) * 8;
        if (MASK && (v * LPR + j) * 8 >= D) continue;
        *reinterpret_cast<bf16x8*>(dZ + off) = pack8(o);
        if (DROP) {
          const uint32_t keep = dropout_keep8(rng, (uint64_t)off >> 3, thresh);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((keep >> e) & 1u) ? o[e] * dscale : 0.f;
          *reinterpret_cast<bf16x8*>(dY + off) = pack8(o);
        }
      }
    }
  }
  // block reduction of the column sums
  const int rslot = wave * RPW + rsub;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + j) * 8;
    if (MASK && c >= D) continue;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[rslot * 2 * D + c + e] = ag[v][e];
      red[rslot * 2 * D + D + c + e] = ab[v][e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += LN_THREADS) {
    float s = 0.f;
    for (int r = 0; r < RPB; ++r) s += red[r * 2 * D + c];
    partial[(long)blockIdx.x * 2 * D + c] = s;
  }
}

// column sums of the per-block partials: 32 columns x 32 row-slices per block (1024 threads), coalesced 128 B rows
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int D,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate) {
  __shared__ float sm[32][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < 2 * D) {
#pragma unroll 4
    for (int b = sl; b < nblk; b += 32) s += partial[(long)b * 2 * D + c];
  }
  sm[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][cl];
    float* dst = c < D ? dgamma + c : dbeta + (c - D);
    *dst = accumulate ? *dst + t : t;
  }
}

struct LnShape { int lpr, nv; };
inline bool ln_shape(int D, LnShape* s) {
  if (D <= 0 || D % 8) return false;
  int v = D / 8, lpr = 1;
  while (lpr < 64 && v % (lpr * 2) == 0) lpr *= 2;
  s->lpr = lpr; s->nv = v / lpr;
  return true;
}

// (LPR, NV) instantiations.  NV is odd below LPR 64 by construction.
template <typename F>
inline bool ln_dispatch(const LnShape& s, F&& f) {
#define IQ_LN_CASE(L, N) \
  if (s.lpr == L && s.nv == N) { f(std::integral_constant<int, L>{}, std::integral_constant<int, N>{}); return true; }
  IQ_LN_CASE(64, 1) IQ_LN_CASE(64, 2) IQ_LN_CASE(64, 3) IQ_LN_CASE(64, 4)
  IQ_LN_CASE(32, 1) IQ_LN_CASE(32, 3) IQ_LN_CASE(32, 5)
  IQ_LN_CASE(16, 1) IQ_LN_CASE(16, 3) IQ_LN_CASE(16, 5)
  IQ_LN_CASE(8, 1) IQ_LN_CASE(8, 3) IQ_LN_CASE(8, 5)
  IQ_LN_CASE(4, 1) IQ_LN_CASE(4, 3) IQ_LN_CASE(4, 5)
  IQ_LN_CASE(2, 1) IQ_LN_CASE(2, 3) IQ_LN_CASE(1, 1) IQ_LN_CASE(1, 3)
#undef IQ_LN_CASE
  return false;
}
// any other D % 8 == 0 up to 2048: one wave per row, lanes past D/8 vectors masked off (slower, always correct)
inline int ln_masked_nv(int D) { return (D % 8 == 0 && D > 0 && D <= 2048) ? (D / 8 + 63) / 64 : 0; }
template <typename F>
inline bool ln_dispatch_masked(int D, F&& f) {
  switch (ln_masked_nv(D)) {
    case 1: f(std::integral_constant<int, 1>{}); return true;
    case 2: f(std::integral_constant<int, 2>{}); return true;
    case 3: f(std::integral_constant<int, 3>{}); return true;
    case 4: f(std::integral_constant<int, 4>{}); return true;
    default: return false;
  }
}

constexpr int LN_MAX_BLOCKS = 512;      // backward: one partial row per block
constexpr int LN_FWD_MAX_BLOCKS = 4096;

inline int ln_grid(int M, int rpb, int cap = LN_MAX_BLOCKS) {
  long nb = ((long)M + rpb - 1) / rpb;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace

extern "C" int iq_ln_supported(int D) {
  LnShape s;
  if (!ln_shape(D, &s)) return 0;
  return (ln_dispatch(s, [](auto, auto) {}) || ln_masked_nv(D) > 0) ? 1 : 0;
}

extern "C" int iq_ln_fwd(const void* z, const float* gamma, const float* beta, void* x, float* mean, float* rstd,
                         int M, int D, float eps, iq_stream_t stream) {
  LnShape s;
  if (M <= 0) return IQ_OK;
  if (!z || !gamma || !beta || !x || !mean || !rstd) return IQ_ERR_ARG;
  if (!ln_shape(D, &s)) return IQ_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_LN_FWD, st);
  IQ_PROF_K(2.0 * (double)M * D * 2 + 8.0 * M, 0.0, "ln_fwd_kernel(D=%d)", D);
  bool ok = ln_dispatch(s, [&](auto lpr, auto nv) {
    constexpr int LPR = decltype(lpr)::value, NV = decltype(nv)::value;
    ln_fwd_kernel<LPR, NV><<<ln_grid(M, (64 / LPR) * 4, LN_FWD_MAX_BLOCKS), LN_THREADS, 0, st>>>((const bf16*)z, gamma, beta, (bf16*)x, mean,
                                                                             rstd, M, D, eps);
  });
  if (!ok) ok = ln_dispatch_masked(D, [&](auto nv) {
    constexpr int NV = decltype(nv)::value;
    ln_fwd_kernel<64, NV, true><<<ln_grid(M, 4, LN_FWD_MAX_BLOCKS), LN_THREADS, 0, st>>>((const bf16*)z, gamma, beta, (bf16*)x,
                                                                                         mean, rstd, M, D, eps);
  });
  return ok ? iq_launch_status() : IQ_ERR_UNSUPPORTED;
}

extern "C" size_t iq_ln_bwd_ws_bytes(int D) { return (size_t)LN_MAX_BLOCKS * 2 * D * sizeof(float); }

extern "C" int iq_ln_bwd_partial_rows(int M, int D) {
  LnShape s;
  if (M <= 0 || !ln_shape(D, &s)) return 0;
  int rows = 0;
  bool ok = ln_dispatch(s, [&](auto lpr, auto nv) {
    constexpr int LPR = decltype(lpr)::value;
    rows = ln_grid(M, (64 / LPR) * 4);
  });
  if (!ok) rows = ln_grid(M, 4);
  return rows;
}

extern "C" int iq_ln_bwd(const void* dx, const void* z, const float* mean, const float* rstd, const float* gamma,
                         void* dz, void* dy, const iq_dropout_t* drop, float* dgamma, float* dbeta, float* ws,
                         int accumulate, int M, int D, iq_stream_t stream) {
  LnShape s;
  if (M <= 0) return IQ_OK;
  if (!dx || !z || !mean || !rstd || !gamma || !dz || !ws) return IQ_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return IQ_ERR_ARG;
  const bool reduce_now = dgamma != nullptr;   // else: partial rows stay in ws for a fused reduction (iq_reduce_seg_t)
  if (!ln_shape(D, &s)) return IQ_ERR_UNSUPPORTED;
  const bool dropping = drop && drop->p > 0.f;
  if (dropping && !dy) return IQ_ERR_ARG;
  IqRng rng = {0, 0, 0, nullptr};
  uint32_t thresh = 0;
  float dscale = 1.f;
  if (dropping) {
    rng.seed = drop->seed; rng.step = drop->step; rng.site = drop->site; rng.step_dev = drop->step_dev;
    thresh = dropout_thresh(drop->p);
    dscale = dropout_scale(drop->p);
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_LN_BWD, st);
  IQ_PROF_K(2.0 * (double)M * D * (3 + (dropping ? 1 : 0)) + 8.0 * M, 0.0, "ln_bwd_kernel(D=%d)", D);
  int rc = IQ_OK;
  bool ok = ln_dispatch(s, [&](auto lpr, auto nv) {
    constexpr int LPR = decltype(lpr)::value, NV = decltype(nv)::value;
    constexpr int RPB = (64 / LPR) * 4;
    const int nblk = ln_grid(M, RPB);
    const size_t lds = (size_t)RPB * 2 * D * sizeof(float);
    if (lds > 160 * 1024) { rc = IQ_ERR_UNSUPPORTED; return; }
    if (dropping) {
      auto k = ln_bwd_kernel<LPR, NV, true>;
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      k<<<nblk, LN_THREADS, lds, st>>>((const bf16*)dx, (const bf16*)z, mean, rstd, gamma, (bf16*)dz, (bf16*)dy, rng,
                                       thresh, dscale, ws, M, D);
    } else {
      auto k = ln_bwd_kernel<LPR, NV, false>;
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      k<<<nblk, LN_THREADS, lds, st>>>((const bf16*)dx, (const bf16*)z, mean, rstd, gamma, (bf16*)dz, (bf16*)dy, rng,
                                       thresh, dscale, ws, M, D);
    }
    if (reduce_now) ln_bwd_reduce_kernel<<<(2 * D + 31) / 32, 1024, 0, st>>>(ws, nblk, D, dgamma, dbeta, accumulate);
  });
  if (!ok) ok = ln_dispatch_masked(D, [&](auto nv) {
    constexpr int NV = decltype(nv)::value;
    constexpr int RPB = 4;
    const int nblk = ln_grid(M, RPB);
    const size_t lds = (size_t)RPB * 2 * D * sizeof(float);
    if (dropping)
      ln_bwd_kernel<64, NV, true, true><<<nblk, LN_THREADS, lds, st>>>((const bf16*)dx, (const bf16*)z, mean, rstd, gamma,
                                                                      (bf16*)dz, (bf16*)dy, rng, thresh, dscale, ws, M, D);
    else
      ln_bwd_kernel<64, NV, false, true><<<nblk, LN_THREADS, lds, st>>>((const bf16*)dx, (const bf16*)z, mean, rstd, gamma,
                                                                       (bf16*)dz, (bf16*)dy, rng, thresh, dscale, ws, M, D);
    if (reduce_now) ln_bwd_reduce_kernel<<<(2 * D + 31) / 32, 1024, 0, st>>>(ws, nblk, D, dgamma, dbeta, accumulate);
  });
  if (!ok) return IQ_ERR_UNSUPPORTED;
  return rc != IQ_OK ? rc : iq_launch_status();
}
